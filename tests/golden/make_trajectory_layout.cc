// Generator of tests/golden/trajectory_layout_expected.json: the STRUCTURE of the reference's trajectory file --
// key order, indentation, one array element per line, the integer 0 of the first angular_velocity, null for a non-finite
// number -- pinned to a REAL nlohmann::json, the 3.1.1 header that sits in the build image (/opt/conda/include/json.hpp;
// the reference builds against 2.1.1, docker/Dockerfile:34).  Objects are std::map in both versions and dump(2)'s layout
// did not change between them; what did change is how a double is printed (2.1.1: "%.15g", 3.x: shortest round trip), so
// the fixture only holds values whose two renderings coincide (integers, dyadic fractions with few digits, x.0, null).
// The object is assembled exactly as src/io/json_converters.cc does it: PoseToJson (:6-18), SetPlane (:37-43),
// SetTrajectory (:56-96), written with `<< dump(2) << std::endl` (src/slam/track_image_sequence.cc:108-109); key strings
// from include/io/json_converters.hpp:10-35.  Input = tests/golden/trajectory_layout_in.txt (the CLI's --trajectory_in
// format).  Build + run:  g++ -std=c++11 -I/opt/conda/include tests/golden/make_trajectory_layout.cc -o /tmp/mtl &&
//   /tmp/mtl tests/golden/trajectory_layout_in.txt > tests/golden/trajectory_layout_expected.json
#include <json.hpp>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1]);
    auto num = [&](double* v) { std::string tok; if (!(f >> tok)) return false; char* e = nullptr; *v = strtod(tok.c_str(), &e); return e != tok.c_str(); };
    double plane[6];
    for (double& v : plane) num(&v);
    nlohmann::json root;
    root["plane"] = {{plane[0], plane[1], plane[2]}, {plane[3], plane[4], plane[5]}};          // SetPlane
    root["trajectory"] = {};
    long long prev_t = 0;
    for (size_t idx = 0;; idx++) {
        long long t, id; int lost; double tr[3], q[4], dx, dy, turn;
        if (!(f >> t >> lost >> id) || !num(&tr[0]) || !num(&tr[1]) || !num(&tr[2]) || !num(&q[0]) || !num(&q[1]) || !num(&q[2]) ||
            !num(&q[3]) || !num(&dx) || !num(&dy) || !num(&turn)) break;
        nlohmann::json point, pose, rotation;
        point["time_usec"] = (int64_t)t;
        point["is_lost"] = lost != 0;
        point["frame_id"] = (int)id;                                                            // frame_id - frame_id_offset (0)
        pose["translation"] = {tr[0], tr[1], tr[2]};
        rotation["w"] = q[0]; rotation["x"] = q[1]; rotation["y"] = q[2]; rotation["z"] = q[3];
        pose["rotation"] = rotation;
        point["pose"] = pose;
        point["planar_direction"] = {dx, dy};
        if (idx == 0) point["angular_velocity"] = 0;                                            // :83
        else point["angular_velocity"] = turn / ((double)(t - prev_t) * 1e-6 + 1e-10);          // :85-90
        prev_t = t;
        root["trajectory"].push_back(point);
    }
    std::cout << root.dump(2) << std::endl;
    return 0;
}
