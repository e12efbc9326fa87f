// trajectory_json.hpp -- trajectory JSON of optical_trajectories, without nlohmann/json.
//
// Reproduces the schema and layout of pilotguru's writer:
//   src/io/json_converters.cc:6-18 (PoseToJson), :37-43 (SetPlane), :56-96 (SetTrajectory),
//   key constants include/io/json_converters.hpp:10-35, dumped with nlohmann::json::dump(2)
//   (src/slam/track_image_sequence.cc:101-109).
// nlohmann/json 2.1.1 (docker/Dockerfile:34) keeps objects in a std::map, so keys come out
// sorted; dump(2) = 2-space indent, one array element per line; floating-point numbers are
// printed with "%.15g" and get ".0" appended when the text has no '.', 'e' or 'E'
// (UNVERIFIED-RECALL of 2.1.1's formatting; pinned by tests/golden/trajectory_expected.json).
// The first trajectory point carries the INTEGER 0 as angular_velocity (json_converters.cc:83).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <string>
#include <vector>

namespace pgorb {

struct Pose { double translation[3]; double qw, qx, qy, qz; };          // ORB_SLAM2::Pose (System.h:46-49)
struct PoseWithTimestamp { Pose pose; int64_t time_usec; bool is_lost; int64_t frame_id; };   // System.h:51-56

inline std::string json_double(double x)
{
    // nlohmann 2.x turns a non-finite float into JSON null when the value is constructed, so the reference's
    // files stay parseable after a diverged fit or a 0/0 angular velocity
    if (!std::isfinite(x)) return "null";
    char buf[64];
    snprintf(buf, sizeof(buf), "%.15g", x);
    std::string s(buf);
    if (s.find_first_of(".eE") == std::string::npos) s += ".0";
    return s;
}

// plane: 2x3 row-major; projected_directions: 2 doubles per point or nullptr; turn_angles: per point or nullptr
inline std::string trajectory_to_json(const double plane[6], const std::vector<PoseWithTimestamp>& trajectory,
                                      const double* projected_directions, const double* turn_angles,
                                      int64_t frame_id_offset)
{
    std::string o = "{\n  \"plane\": [\n";
    for (int r = 0; r < 2; r++) {
        o += "    [\n";
        for (int c = 0; c < 3; c++) o += "      " + json_double(plane[3 * r + c]) + (c < 2 ? ",\n" : "\n");
        o += std::string("    ]") + (r == 0 ? ",\n" : "\n");
    }
    o += "  ],\n  \"trajectory\": ";
    if (trajectory.empty()) { o += "null\n}"; return o; }                // (*json_root)[kTrajectory] = {} stays null
    o += "[\n";
    for (size_t i = 0; i < trajectory.size(); i++) {
        const PoseWithTimestamp& p = trajectory[i];
        o += "    {\n";
        if (turn_angles) {
            if (i == 0) o += "      \"angular_velocity\": 0,\n";
            else {
                const double dt = static_cast<double>(p.time_usec - trajectory[i - 1].time_usec) * 1e-6;
                o += "      \"angular_velocity\": " + json_double(turn_angles[i] / (dt + 1e-10)) + ",\n";
            }
        }
        o += "      \"frame_id\": " + std::to_string((long long)(p.frame_id - frame_id_offset)) + ",\n";
        o += std::string("      \"is_lost\": ") + (p.is_lost ? "true" : "false") + ",\n";
        if (projected_directions)
            o += "      \"planar_direction\": [\n        " + json_double(projected_directions[2 * i]) + ",\n        " +
                 json_double(projected_directions[2 * i + 1]) + "\n      ],\n";
        o += "      \"pose\": {\n        \"rotation\": {\n";
        o += "          \"w\": " + json_double(p.pose.qw) + ",\n          \"x\": " + json_double(p.pose.qx) + ",\n";
        o += "          \"y\": " + json_double(p.pose.qy) + ",\n          \"z\": " + json_double(p.pose.qz) + "\n        },\n";
        o += "        \"translation\": [\n          " + json_double(p.pose.translation[0]) + ",\n          " +
             json_double(p.pose.translation[1]) + ",\n          " + json_double(p.pose.translation[2]) + "\n        ]\n      },\n";
        o += "      \"time_usec\": " + std::to_string((long long)p.time_usec) + "\n";
        o += std::string("    }") + (i + 1 < trajectory.size() ? ",\n" : "\n");
    }
    o += "  ]\n}";
    return o;
}

}  // namespace pgorb
