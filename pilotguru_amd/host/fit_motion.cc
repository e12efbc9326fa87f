// fit_motion.cc -- the drop-in CLI surface of pilotguru's fit_motion (src/fit_motion.cc) over libpgorb.
//
// Same flags, CHECKs, inputs and outputs as the reference:
//   --rotations_json / --accelerations_json / --locations_json       PilotGuru Recorder raw data (:45-60)
//   --velocities_out_json / --steering_out_json / --forward_axis_out_json   (:64-83)
//   --locations_batch_size=40 --locations_shift_step=5 --optimization_iters=500 --post_smoothing_sigma_sec=0.003
//   --principal_rotation_axis_integration_interval_usec=500000
//   --forward_axis_inference_min_velocity_m_s=5.0 --forward_axis_inference_min_rotation_rad=0.2   (:87-113)
// What runs: GetPrincipalRotationAxes -> vertical axis (:322-329), the steering series (:130-148), and
// ComputeAndSaveForwardVelocitiesFromImu (:151-290) with every sliding-window L-BFGS fit on the GPU
// (pgorb_fit_motion_velocities, csrc/calib.hip).  JSON files are written like nlohmann::json::dump(2)
// writes them (keys sorted, see trajectory_json.hpp).  --device picks the GPU.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../include/pgorb.h"
#include "mini_json.hpp"
#include "trajectory_json.hpp"

namespace {

[[noreturn]] void check_failed(const char* what)
{
    fprintf(stderr, "Check failed: %s\n", what);
    exit(EXIT_FAILURE);
}

struct Flags {
    std::string rotations_json, accelerations_json, locations_json, velocities_out_json, steering_out_json, forward_axis_out_json;
    long long locations_batch_size = 40, locations_shift_step = 5, optimization_iters = 500;
    long long principal_rotation_axis_integration_interval_usec = 500000;
    double post_smoothing_sigma_sec = 0.003, forward_axis_inference_min_velocity_m_s = 5.0, forward_axis_inference_min_rotation_rad = 0.2;
    int device = 0;
};

bool parse_flags(int argc, char** argv, Flags& F)
{
    std::map<std::string, std::string*> str = {{"rotations_json", &F.rotations_json}, {"accelerations_json", &F.accelerations_json},
        {"locations_json", &F.locations_json}, {"velocities_out_json", &F.velocities_out_json}, {"steering_out_json", &F.steering_out_json},
        {"forward_axis_out_json", &F.forward_axis_out_json}};
    std::map<std::string, long long*> ints = {{"locations_batch_size", &F.locations_batch_size}, {"locations_shift_step", &F.locations_shift_step},
        {"optimization_iters", &F.optimization_iters},
        {"principal_rotation_axis_integration_interval_usec", &F.principal_rotation_axis_integration_interval_usec}};
    std::map<std::string, double*> reals = {{"post_smoothing_sigma_sec", &F.post_smoothing_sigma_sec},
        {"forward_axis_inference_min_velocity_m_s", &F.forward_axis_inference_min_velocity_m_s},
        {"forward_axis_inference_min_rotation_rad", &F.forward_axis_inference_min_rotation_rad}};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0) a = a.substr(2); else if (a.rfind("-", 0) == 0) a = a.substr(1); else return false;
        std::string name = a, val;
        const size_t eq = a.find('=');
        if (eq != std::string::npos) { name = a.substr(0, eq); val = a.substr(eq + 1); }
        else { if (i + 1 >= argc) return false; val = argv[++i]; }
        if (str.count(name)) *str[name] = val;
        else if (ints.count(name)) *ints[name] = atoll(val.c_str());
        else if (reals.count(name)) *reals[name] = strtod(val.c_str(), nullptr);
        else if (name == "device") F.device = atoi(val.c_str());
        else { fprintf(stderr, "ERROR: unknown command line flag '%s'\n", name.c_str()); return false; }
    }
    return true;
}

// ReadTimestamp3DData (:116-129): {field: [{"x":..,"y":..,"z":..,"time_usec":..}, ...]}
void read_3d(const std::string& file, const char* field, std::vector<double>& v, std::vector<int64_t>& t)
{
    pgorb::JsonValue root;
    if (!pgorb::read_json_file(file, root)) check_failed("input JSON parses");
    const pgorb::JsonValue& list = root.at(field);
    if (list.kind != pgorb::JsonValue::Array || list.a.empty()) check_failed("!entries_list.empty()");      // :120
    for (const pgorb::JsonValue& e : list.a) {
        if (!e.at("x").is_number() || !e.at("y").is_number() || !e.at("z").is_number() || !e.at("time_usec").is_number())
            check_failed("entry has numeric x, y, z, time_usec");
        v.push_back(e.at("x").as_double()); v.push_back(e.at("y").as_double()); v.push_back(e.at("z").as_double());
        t.push_back(e.at("time_usec").as_int());
    }
}

// ReadGpsVelocities (:131-144): {"locations": [{"speed_m_s":.., "time_usec":.., ...}, ...]}
void read_gps(const std::string& file, std::vector<double>& v, std::vector<int64_t>& t)
{
    pgorb::JsonValue root;
    if (!pgorb::read_json_file(file, root)) check_failed("input JSON parses");
    const pgorb::JsonValue& list = root.at("locations");
    if (list.kind != pgorb::JsonValue::Array || list.a.empty()) check_failed("!locations_json.empty()");    // :135
    for (const pgorb::JsonValue& e : list.a) {
        if (!e.at("speed_m_s").is_number() || !e.at("time_usec").is_number()) check_failed("location has numeric speed_m_s, time_usec");
        v.push_back(e.at("speed_m_s").as_double());
        t.push_back(e.at("time_usec").as_int());
    }
}

// JsonWriteTimestampedRealData (src/io/json_converters.cc:184-202) + dump(2) + endl
void write_timestamped(const std::vector<int64_t>& t, const double* v, const std::string& file, const char* root, const char* value_name)
{
    std::string o = std::string("{\n  \"") + root + "\": ";
    if (t.empty()) o += "null\n}";                                        // out_json[root] = {} stays null without a push_back
    else {
        o += "[\n";
        const bool value_first = std::string(value_name) < "time_usec";   // std::map order of the two keys
        for (size_t i = 0; i < t.size(); i++) {
            const std::string a = std::string("      \"") + value_name + "\": " + pgorb::json_double(v[i]);
            const std::string b = "      \"time_usec\": " + std::to_string((long long)t[i]);
            o += "    {\n" + (value_first ? a + ",\n" + b : b + ",\n" + a) + "\n    }" + (i + 1 < t.size() ? ",\n" : "\n");
        }
        o += "  ]\n}";
    }
    std::ofstream f(file);
    if (!f.good()) check_failed("output JSON is writable");
    f << o << std::endl;
}

}  // namespace

int main(int argc, char** argv)
{
    Flags F;
    if (!parse_flags(argc, argv, F)) return EXIT_FAILURE;
    if (F.rotations_json.empty()) check_failed("!FLAGS_rotations_json.empty()");                       // :298-306
    if (F.accelerations_json.empty()) check_failed("!FLAGS_accelerations_json.empty()");
    if (F.locations_json.empty()) check_failed("!FLAGS_locations_json.empty()");
    if (!(F.optimization_iters > 0)) check_failed("FLAGS_optimization_iters > 0");
    if (!(F.locations_batch_size > 0)) check_failed("FLAGS_locations_batch_size > 0");
    if (!(F.locations_shift_step > 0)) check_failed("FLAGS_locations_shift_step > 0");
    if (!(F.locations_batch_size >= F.locations_shift_step)) check_failed("FLAGS_locations_batch_size >= FLAGS_locations_shift_step");
    if (!(F.post_smoothing_sigma_sec > 0)) check_failed("FLAGS_post_smoothing_sigma_sec > 0");
    if (!(F.principal_rotation_axis_integration_interval_usec > 0)) check_failed("FLAGS_principal_rotation_axis_integration_interval_usec > 0");

    std::vector<double> gps_v, rot, acc;
    std::vector<int64_t> gps_t, rot_t, acc_t;
    read_gps(F.locations_json, gps_v, gps_t);
    read_3d(F.rotations_json, "rotations", rot, rot_t);
    read_3d(F.accelerations_json, "accelerations", acc, acc_t);

    double axes[9];
    if (pgorb_principal_rotation_axes(rot.data(), rot_t.data(), (int)rot_t.size(), F.principal_rotation_axis_integration_interval_usec, axes) != PGORB_OK)
        check_failed("interval_rotations.size() >= 3");                                               // rotation.cc:47
    const double* vertical_axis = axes;                                                               // row 0 (:324-329)

    if (!F.steering_out_json.empty()) {                                                               // :331-334, :130-148
        std::vector<double> steering(rot_t.size());
        if (pgorb_angular_velocities_around_axis(rot.data(), (int)rot_t.size(), vertical_axis, steering.data()) != PGORB_OK)
            check_failed("axis_norm within 1e-2 of 1");
        write_timestamped(rot_t, steering.data(), F.steering_out_json, "steering", "angular_velocity");
    }

    if (!F.velocities_out_json.empty() || !F.forward_axis_out_json.empty()) {                         // :336-347
        pgorb_params prm = {};
        prm.nfeatures = 500; prm.scale_factor = 1.2f; prm.nlevels = 4; prm.ini_th_fast = 20; prm.min_th_fast = 7;
        prm.max_width = 320; prm.max_height = 240; prm.max_batch = 1; prm.device = F.device;
        pgorb_ctx* ctx = nullptr;
        if (pgorb_create(&prm, &ctx) != PGORB_OK) check_failed("a gfx950 device is present (there is no CPU fallback)");
        std::vector<int64_t> out_t(rot_t.size() + acc_t.size());
        std::vector<double> out_v(out_t.size());
        int n = 0;
        double forward_axis[3];
        if (pgorb_fit_motion_velocities(ctx, gps_v.data(), gps_t.data(), (int)gps_t.size(), rot.data(), rot_t.data(), (int)rot_t.size(),
                                        acc.data(), acc_t.data(), (int)acc_t.size(), vertical_axis, (int)F.locations_batch_size,
                                        (int)F.locations_shift_step, (int)F.optimization_iters, F.post_smoothing_sigma_sec,
                                        F.forward_axis_inference_min_velocity_m_s, F.forward_axis_inference_min_rotation_rad,
                                        out_t.data(), out_v.data(), &n, forward_axis) != PGORB_OK)
            check_failed(pgorb_last_error(ctx));
        out_t.resize(n);
        if (!F.velocities_out_json.empty()) write_timestamped(out_t, out_v.data(), F.velocities_out_json, "velocities", "speed_m_s");
        if (!F.forward_axis_out_json.empty()) {                                                       // :279-288
            std::ofstream f(F.forward_axis_out_json);
            if (!f.good()) check_failed("output JSON is writable");
            f << "{\n  \"forward_axis\": {\n    \"x\": " << pgorb::json_double(forward_axis[0]) << ",\n    \"y\": " << pgorb::json_double(forward_axis[1])
              << ",\n    \"z\": " << pgorb::json_double(forward_axis[2]) << "\n  }\n}" << std::endl;
        }
        pgorb_destroy(ctx);
    }
    return EXIT_SUCCESS;
}
