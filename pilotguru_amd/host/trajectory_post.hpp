// trajectory_post.hpp -- C++ face of the trajectory post-processing in libpgorb (csrc/post.cc),
// with the reference's function names (src/slam/smoothing.cc, src/slam/horizontal_flatten.cc,
// src/slam/track_image_sequence.cc:16-29) so the tail of TrackImageSequence reads the same.
#pragma once
#include <vector>

#include "../../include/pgorb.h"
#include "trajectory_json.hpp"

namespace pgorb {

inline bool SmoothHeadingDirections(std::vector<PoseWithTimestamp>* trajectory, int sigma)
{
    std::vector<double> q(trajectory->size() * 4);
    for (size_t i = 0; i < trajectory->size(); i++) {
        const Pose& p = (*trajectory)[i].pose;
        q[4 * i] = p.qw; q[4 * i + 1] = p.qx; q[4 * i + 2] = p.qy; q[4 * i + 3] = p.qz;
    }
    if (pgorb_smooth_heading_directions(q.data(), (int)trajectory->size(), sigma) != PGORB_OK) return false;
    for (size_t i = 0; i < trajectory->size(); i++) {
        Pose& p = (*trajectory)[i].pose;
        p.qw = q[4 * i]; p.qx = q[4 * i + 1]; p.qy = q[4 * i + 2]; p.qz = q[4 * i + 3];
    }
    return true;
}

struct TrajectoryPCA { double eigenvectors[9], eigenvalues[3], mean[3]; };

inline bool TrajectoryToPCA(const std::vector<PoseWithTimestamp>& trajectory, TrajectoryPCA* pca)
{
    std::vector<double> t(trajectory.size() * 3);
    for (size_t i = 0; i < trajectory.size(); i++) for (int c = 0; c < 3; c++) t[3 * i + c] = trajectory[i].pose.translation[c];
    return pgorb_trajectory_pca(t.data(), (int)trajectory.size(), pca->eigenvectors, pca->eigenvalues, pca->mean) == PGORB_OK;
}

inline std::vector<double> ProjectDirections(const std::vector<PoseWithTimestamp>& trajectory, const double plane[6])
{
    std::vector<double> q(trajectory.size() * 4), dirs(trajectory.size() * 2);
    for (size_t i = 0; i < trajectory.size(); i++) {
        const Pose& p = trajectory[i].pose;
        q[4 * i] = p.qw; q[4 * i + 1] = p.qx; q[4 * i + 2] = p.qy; q[4 * i + 3] = p.qz;
    }
    pgorb_project_directions(q.data(), (int)trajectory.size(), plane, dirs.data());
    return dirs;
}

inline std::vector<double> Projected2DDirectionsToTurnAngles(const std::vector<double>& directions)
{
    std::vector<double> turn(directions.size() / 2);
    pgorb_turn_angles(directions.data(), (int)turn.size(), turn.data());
    return turn;
}

}  // namespace pgorb
