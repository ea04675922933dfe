// BundleAdjusterHip.hpp — C++ adapter that restates eacham's RefineBA on top of the C-ABI.
//
//   void RefineBA(const int currentFrameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map,
//                 cv::Mat& K, const OptimizerConfig& config);
//       /root/reference/modules/sfm/reconstruction/BundleAdjuster.h:13-17, .cpp:40-250
//   struct OptimizerConfig { method, maxIter, maxTolerance, delta, usePreconditioner }
//       /root/reference/modules/sfm/config/SfmConfig.h:15-22
//
// OpenCV / Eigen are not required: the graph and the map are seen through two small views that a
// caller fills from eacham's Graph/Node/Map accessors (INTEGRATION.md shows the ten lines that do it).
// The adapter performs the reference's graph walk — window selection (:123-162), the landmark filter
// `status && observers >= 2` (:84), first-seen landmark registration (:100-117) — hands plain arrays
// to eacham_ba_solve, and writes the result back (:221-249).
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../eacham_hip.h"
#include "FlatMap.hpp"

namespace eacham {
namespace hip {

struct OptimizerConfig {  // fields verbatim
    std::string method = "LM";
    int maxIter = 100;
    float maxTolerance = 1e-5f;
    float delta = 10.0f;
    bool usePreconditioner = false;
};

// What RefineBA reads from / writes to one Node (modules/sfm/data/Node.h).
struct NodeView {
    unsigned id = 0;
    bool valid = false;                                // Node::IsValid()
    bool fixed = false;                                // Graph::IsFixed(id)
    double transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // Node::GetTransform(), row-major world->camera
    std::vector<float> keypoints;                      // x0 y0 x1 y1 ... (Node::GetKeyPoint(id2d), cv::Point2f)
    FlatMap points3d;                                  // Node::GetPoints3d(): keypoint index -> landmark id (ascending keypoint order)
    std::vector<unsigned> neighbours;                  // keys of Node::GetFactors()
};

// What RefineBA reads from / writes to one map point (modules/sfm/data/Map.h:15-23).
struct MapPointView {
    double point3d[3] = {0, 0, 0};
    bool status = false;        // Map::GetStatus(id)
    unsigned observers = 0;     // Map::GetObservers(id).size()
};

struct GraphView {
    std::map<unsigned, NodeView> nodes;  // Graph::GetNodes()
};
struct MapView {
    std::unordered_map<unsigned, MapPointView> points;
};

struct RefineBAReport {
    bool skipped = false;  // fewer than 50 landmarks: the reference returns silently (:166-169)
    double initial_error = 0, final_error = 0;
    int outer_iterations = 0, inner_iterations = 0;
    size_t frames = 0, landmarks = 0, observations = 0;
};

// K: 3x3 row-major camera matrix (cv::Mat CV_64F in the reference); entries (0,0) (1,1) (0,2) (1,2).
inline RefineBAReport RefineBA(eacham_ctx* ctx, const int currentFrameId, GraphView& graph, MapView& map, double* K,
                               const OptimizerConfig& config) {
    std::vector<unsigned> frameIds;
    std::vector<double> camT;
    std::vector<int32_t> camFixed, pointObservers;
    std::vector<double> points, uv;
    std::vector<uint32_t> obsCam, obsPoint;
    std::vector<unsigned> mapIds;                      // dense point index -> landmark id
    std::unordered_map<unsigned, uint32_t> mapIndex;   // landmark id -> dense point index

    auto frameAdder = [&](const NodeView& node) {      // BundleAdjuster.cpp:57-121
        const uint32_t cam = (uint32_t)frameIds.size();
        frameIds.push_back(node.id);
        camT.insert(camT.end(), node.transform, node.transform + 16);
        camFixed.push_back(node.fixed ? 1 : 0);
        for (const auto& kv : node.points3d) {
            const unsigned id2d = kv.first, id3d = kv.second;
            const auto it = map.points.find(id3d);
            if (it == map.points.end()) throw std::runtime_error("RefineBA: map point not found");
            if (!it->second.status || it->second.observers < 2) continue;           // :84
            if (2 * (size_t)id2d + 1 >= node.keypoints.size()) throw std::runtime_error("RefineBA: keypoint out of range");
            auto ins = mapIndex.insert({id3d, (uint32_t)mapIds.size()});
            if (ins.second) {                                                        // :100-117
                mapIds.push_back(id3d);
                points.insert(points.end(), it->second.point3d, it->second.point3d + 3);
                pointObservers.push_back((int32_t)it->second.observers);
            }
            obsCam.push_back(cam);
            obsPoint.push_back(ins.first->second);
            uv.push_back((double)node.keypoints[2 * id2d]);
            uv.push_back((double)node.keypoints[2 * id2d + 1]);
        }
    };

    if (currentFrameId > -1) {                          // local window (:123-145)
        const auto start = graph.nodes.find((unsigned)currentFrameId);
        if (start == graph.nodes.end()) throw std::runtime_error("Node is null");
        frameAdder(start->second);
        for (unsigned id : start->second.neighbours) {
            const auto it = graph.nodes.find(id);
            if (it == graph.nodes.end()) throw std::runtime_error("Node is null");
            if (it->second.valid) frameAdder(it->second);
        }
    } else {                                            // global (:146-162)
        for (const auto& kv : graph.nodes)
            if (kv.second.valid) frameAdder(kv.second);
    }

    RefineBAReport rep;
    rep.frames = frameIds.size();
    rep.landmarks = mapIds.size();
    rep.observations = obsCam.size();

    eacham_ba_problem prob{};
    prob.n_cams = (int32_t)frameIds.size();
    prob.n_points = (int32_t)mapIds.size();
    prob.n_obs = (int32_t)obsCam.size();
    prob.cam_T_wc = camT.data();
    prob.cam_fixed = camFixed.data();
    prob.points = points.data();
    prob.point_observers = pointObservers.data();
    prob.obs_cam = obsCam.data();
    prob.obs_point = obsPoint.data();
    prob.obs_uv = uv.data();
    prob.K[0] = K[0]; prob.K[1] = K[4]; prob.K[2] = K[2]; prob.K[3] = K[5];          // :47-49

    eacham_ba_options opt{};
    if (config.method == "LM") opt.method = EACHAM_BA_LM;
    else if (config.method == "DogLeg") opt.method = EACHAM_BA_DOGLEG;
    else throw std::runtime_error("RefineBA: unknown method " + config.method);      // the reference would dereference a null optimizer
    opt.max_iter = config.maxIter;
    opt.max_tolerance = config.maxTolerance;
    opt.delta = config.delta;
    opt.use_preconditioner = config.usePreconditioner ? 1 : 0;
    opt.min_landmarks = 50;                                                           // :166

    std::vector<double> outT(camT.size()), outP(points.size());
    eacham_ba_result res{};
    res.cam_T_wc = outT.data();
    res.points = outP.data();
    const int rc = eacham_ba_solve(ctx, &prob, &opt, &res);
    if (rc != EACHAM_OK) throw std::runtime_error(std::string("eacham_hip: ") + eacham_last_error(ctx));
    rep.skipped = res.status == EACHAM_BA_SKIPPED;
    if (rep.skipped) return rep;
    rep.initial_error = res.initial_error;
    rep.final_error = res.final_error;
    rep.outer_iterations = res.outer_iterations;
    rep.inner_iterations = res.inner_iterations;

    K[0] = res.K[0]; K[4] = res.K[1]; K[2] = res.K[2]; K[5] = res.K[3];              // :224-227
    for (size_t j = 0; j < mapIds.size(); ++j) {                                     // :229-236
        MapPointView& mp = map.points[mapIds[j]];
        for (int a = 0; a < 3; ++a) mp.point3d[a] = outP[3 * j + a];
        mp.status = true;
    }
    for (size_t i = 0; i < frameIds.size(); ++i) {                                   // :238-248
        NodeView& node = graph.nodes[frameIds[i]];
        for (int a = 0; a < 16; ++a) node.transform[a] = outT[16 * i + a];
    }
    return rep;
}

}  // namespace hip
}  // namespace eacham
