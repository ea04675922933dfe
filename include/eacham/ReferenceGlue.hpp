// ReferenceGlue.hpp — the reference-typed entry points, as compiled code.
//
//   void RefineBA(const int currentFrameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map,
//                 cv::Mat& K, const OptimizerConfig& config);
//       /root/reference/modules/sfm/reconstruction/BundleAdjuster.h:13-17 (call sites apps/sfm/main.cpp:207, :230)
//   void TriangulateFrame(const unsigned frameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map,
//                         const cv::Mat& K, const unsigned minObservers, const float maxReprError,
//                         const float minTriAngle);
//       /root/reference/modules/sfm/reconstruction/Triangulator.h:41-43 (call sites apps/sfm/main.cpp:203, :209)
//
// These are the two functions a maintainer swaps in: same names, same arguments, same effects on Graph / Node / Map /
// K as the reference's (window poses, landmark positions and status, K; new map points, observers, Node::SetPoint3d).
// They fill the POD views of BundleAdjusterHip.hpp / TriangulatorHip.hpp from the accessors of
// modules/sfm/data/{Graph,Node,Map}.h, call the adapters (which call the C-ABI) and write the result back.
//
// The bodies are templates over the graph / map / matrix types: they only use the reference's accessor NAMES, so the
// same code is compiled (1) against eacham's own headers when they and OpenCV / Eigen are on the include path — the
// overloads at the bottom then have exactly the reference's signatures — and (2) in this repository's test-suite against
// minimal stand-ins with those accessor names (tests/cpp/ref_standins.hpp; OpenCV, Eigen and the reference headers are
// not in the image), driven by tests/cpp/adapter_driver.cpp / tri_driver.cpp with -DEACHAM_TEST_GLUE.
#pragma once

#include <memory>
#ifdef EACHAM_GLUE_TIMING
#include <chrono>
#include <cstdio>
#endif
#include <set>
#include <stdexcept>
#include <type_traits>
#include <utility>

#include "BundleAdjusterHip.hpp"
#include "TriangulatorHip.hpp"

#if !defined(EACHAM_GLUE_STANDINS) && defined(__has_include)
#if __has_include(<opencv4/opencv2/core.hpp>) && __has_include(<Eigen/Core>) && __has_include("sfm/data/Graph.h") && \
    __has_include("sfm/data/Map.h") && __has_include("sfm/config/SfmConfig.h")
#include <opencv4/opencv2/core.hpp>
#include <Eigen/Core>
#include "sfm/data/Graph.h"
#include "sfm/data/Map.h"
#include "sfm/config/SfmConfig.h"
#define EACHAM_GLUE_REFERENCE_TYPES 1
#endif
#endif

namespace eacham {
namespace hip {
namespace glue {

inline Context& shared_context() {  // one device context for the process, as the reference has one matcher / one optimiser
    static Context ctx;
    return ctx;
}

// world->camera 4x4 of a node as 16 row-major doubles (Eigen::Matrix4d is column-major: element access only)
template <class Mat4>
inline void matrix_to_rows(const Mat4& M, double* out) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = M(r, c);
}

// RefineBA with the reference's argument list (BundleAdjuster.cpp:40-250).
template <class GraphT, class MapT, class MatT, class ConfigT>
inline RefineBAReport RefineBA(const int currentFrameId, const std::shared_ptr<GraphT>& graph, const std::shared_ptr<MapT>& map,
                               MatT& K, const ConfigT& config) {
    GraphView gv;
    MapView mv;
    // Only what the adapter reads is converted: the local window is the current frame and its factor neighbours
    // (BundleAdjuster.cpp:123-145), the global problem (currentFrameId = -1) every node. Converting the whole graph and map per
    // call made the incremental loop quadratic in the sequence length (7.9 ms per TriangulateFrame at 100 frames).
    std::set<unsigned> needed;
    if (currentFrameId > -1) {
        auto* start = graph->Get((unsigned)currentFrameId);
        if (!start) throw std::runtime_error("Node is null");
        needed.insert((unsigned)currentFrameId);
        for (const auto& f : start->GetFactors()) needed.insert(f.first);
    }
    for (const auto& entry : graph->GetNodes()) {
        const unsigned id = entry.first;
        if (currentFrameId > -1 && !needed.count(id)) continue;
        auto* node = entry.second;
        NodeView nv;
        nv.id = id;
        nv.valid = node->IsValid();
        nv.fixed = graph->IsFixed(id);
        matrix_to_rows(node->GetTransform(), nv.transform);
        nv.keypoints.reserve(2 * node->GetFeatures().size());
        for (const auto& kp : node->GetFeatures()) {   // Node::GetKeyPoint(id2d) of every keypoint (cv::Point2f)
            nv.keypoints.push_back(kp.x);
            nv.keypoints.push_back(kp.y);
        }
        {
            std::vector<FlatMap::value_type> items;
            items.reserve(node->GetPoints3d().size());
            for (const auto& p : node->GetPoints3d()) items.emplace_back(p.first, p.second);
            nv.points3d.assign_unsorted(std::move(items));
        }
        const auto& all = map->GetAll();
        for (const auto& p : nv.points3d) {
            if (mv.points.find(p.second) == mv.points.end()) {
                const auto it = all.find(p.second);   // (Map::Get / GetStatus / GetObservers of the reference each take a lock and the last one copies the map)
                if (it == all.end()) throw std::runtime_error("Map: point is not found");
                MapPointView mp;
                mp.point3d[0] = it->second.point3d(0), mp.point3d[1] = it->second.point3d(1), mp.point3d[2] = it->second.point3d(2);
                mp.status = it->second.isValid;
                mp.observers = (unsigned)it->second.observers.size();
                mv.points[p.second] = mp;
            }
        }
        for (const auto& f : node->GetFactors()) nv.neighbours.push_back(f.first);
        gv.nodes[id] = std::move(nv);
    }
    double K9[9] = {K.template at<double>(0, 0), 0.0, K.template at<double>(0, 2), 0.0, K.template at<double>(1, 1),
                    K.template at<double>(1, 2), 0.0, 0.0, 1.0};
    OptimizerConfig c;
    c.method = config.method;
    c.maxIter = config.maxIter;
    c.maxTolerance = config.maxTolerance;
    c.delta = config.delta;
    c.usePreconditioner = config.usePreconditioner;
    // what the adapter is about to overwrite: only poses / points that moved are written back
    GraphView before_g = gv;
    MapView before_m = mv;
    const RefineBAReport rep = eacham::hip::RefineBA(shared_context().get(), currentFrameId, gv, mv, K9, c);
    if (rep.skipped) return rep;  // fewer than 50 landmarks: the reference returns without touching anything (:166-169)
    K.template at<double>(0, 0) = K9[0];   // :224-227
    K.template at<double>(1, 1) = K9[4];
    K.template at<double>(0, 2) = K9[2];
    K.template at<double>(1, 2) = K9[5];
    using Vec3 = std::decay_t<decltype(map->Get(0u))>;
    for (const auto& kv : mv.points) {     // :229-241 (every landmark of the problem: UpdatePoint + UpdateStatus(true))
        const MapPointView& was = before_m.points[kv.first];
        const MapPointView& now = kv.second;
        if (now.point3d[0] != was.point3d[0] || now.point3d[1] != was.point3d[1] || now.point3d[2] != was.point3d[2] ||
            now.status != was.status) {
            Vec3 X;
            X(0) = now.point3d[0], X(1) = now.point3d[1], X(2) = now.point3d[2];
            map->UpdatePoint(kv.first, X);
            map->UpdateStatus(kv.first, now.status);
        }
    }
    for (const auto& kv : gv.nodes) {      // :243-248
        const NodeView& was = before_g.nodes[kv.first];
        bool moved = false;
        for (int k = 0; k < 16; ++k) moved = moved || kv.second.transform[k] != was.transform[k];
        if (!moved) continue;
        auto* node = graph->Get(kv.first);
        using Mat4 = std::decay_t<decltype(node->GetTransform())>;
        Mat4 M;
        for (int r = 0; r < 4; ++r)
            for (int col = 0; col < 4; ++col) M(r, col) = kv.second.transform[4 * r + col];
        node->SetTransform(M);
    }
    return rep;
}

// TriangulateFrame with the reference's argument list (Triangulator.cpp:188-300). `color`: the reference draws one per
// call from cv::RNG (:192-199, display only); the caller's colour type is default-constructed here.
template <class GraphT, class MapT, class MatT>
inline TriangulateFrameReport TriangulateFrame(const unsigned frameId, const std::shared_ptr<GraphT>& graph,
                                               const std::shared_ptr<MapT>& map, const MatT& K, const unsigned minObservers,
                                               const float maxReprError, const float minTriAngle) {
#ifdef EACHAM_GLUE_TIMING
    struct Tm { double conv = 0, call = 0, back = 0; long n = 0; ~Tm() { std::fprintf(stderr, "TriangulateFrame glue: %ld calls, convert %.3f ms, adapter %.3f ms, write-back %.3f ms per call\n", n, conv / n, call / n, back / n); } };
    static Tm tm;
    const auto t_a = std::chrono::steady_clock::now();
#endif
    TriGraphView gv;
    TriMapView mv;
    // the walk reads the frame, the nodes it has factors to, and the map points those nodes reference (Triangulator.cpp:204-296)
    std::set<unsigned> needed{frameId};
    {
        auto* start = graph->Get(frameId);
        if (!start) throw std::runtime_error("Node is null");
        for (const auto& f : start->GetFactors()) needed.insert(f.first);
    }
    for (const auto& entry : graph->GetNodes()) {
        if (!needed.count(entry.first)) continue;
        auto* node = entry.second;
        TriNodeView nv;
        nv.valid = node->IsValid();
        matrix_to_rows(node->GetTransform(), nv.transform);
        nv.keypoints.reserve(2 * node->GetFeatures().size());
        for (const auto& kp : node->GetFeatures()) {
            nv.keypoints.push_back(kp.x);
            nv.keypoints.push_back(kp.y);
        }
        {
            std::vector<FlatMap::value_type> items;
            items.reserve(node->GetPoints3d().size());
            for (const auto& p : node->GetPoints3d()) items.emplace_back(p.first, p.second);
            nv.points3d.assign_unsorted(std::move(items));
        }
        if (entry.first == frameId)   // the walk only reads the factors of the frame being inserted (:204)
            for (const auto& f : node->GetFactors()) {
                auto& dst = nv.factors[f.first];
                for (const auto& mm : f.second.matches) dst.emplace_back(mm.first, mm.second);
                std::sort(dst.begin(), dst.end());  // (match_t is an unordered_map: no order to keep)
            }
        gv.nodes[entry.first] = std::move(nv);
    }
    const auto& all = map->GetAll();
    for (const auto& kv : all)
        if (kv.first > mv.mapPointId) mv.mapPointId = kv.first;  // Map never removes a point: its counter is the largest id
    for (const auto& nk : gv.nodes)
        for (const auto& p : nk.second.points3d) {
            if (mv.points.count(p.second)) continue;
            const auto it = all.find(p.second);
            if (it == all.end()) throw std::runtime_error("Map: point is not found");
            TriMapPoint mp;
            mp.point3d[0] = it->second.point3d(0), mp.point3d[1] = it->second.point3d(1), mp.point3d[2] = it->second.point3d(2);
            mp.isValid = it->second.isValid;
            {
                std::vector<FlatMap::value_type> items;
                items.reserve(it->second.observers.size());
                for (const auto& ob : it->second.observers) items.emplace_back(ob.first, ob.second);
                mp.observers.assign_unsorted(std::move(items));
            }
            mv.points[p.second] = std::move(mp);
        }
    const unsigned lastId = mv.mapPointId;   // ids above it after the call are new map points
    const double K9[9] = {K.template at<double>(0, 0), 0.0, K.template at<double>(0, 2), 0.0, K.template at<double>(1, 1),
                          K.template at<double>(1, 2), 0.0, 0.0, 1.0};
#ifdef EACHAM_GLUE_TIMING
    const auto t_b = std::chrono::steady_clock::now();
#endif
    const TriangulateFrameReport rep =
        eacham::hip::TriangulateFrame(shared_context(), frameId, gv, mv, K9, minObservers, maxReprError, minTriAngle);
#ifdef EACHAM_GLUE_TIMING
    const auto t_c = std::chrono::steady_clock::now();
#endif
    // ---- write-back, in the order the reference mutates its objects; the views are compared with the LIVE objects (which the
    //      adapter never touched) instead of with copies taken before the call ----
    using Vec3 = std::decay_t<decltype(map->Get(0u))>;
    for (auto& kv : mv.points) {           // new map points: Map::Add hands out the ids the view counted up
        if (kv.first <= lastId) continue;
        Vec3 X;
        X(0) = kv.second.point3d[0], X(1) = kv.second.point3d[1], X(2) = kv.second.point3d[2];
        const unsigned id = map->Add(X);
        if (id != kv.first) throw std::runtime_error("TriangulateFrame glue: Map::Add returned an unexpected id");
    }
    for (const auto& kv : mv.points) {     // observers and validity
        const bool isNew = kv.first > lastId;
        std::vector<std::pair<unsigned, unsigned>> gone, come;
        bool wasValid = false;
        {
            const auto live = map->GetAll().find(kv.first);
            if (live == map->GetAll().end()) throw std::runtime_error("Map: point is not found");
            wasValid = live->second.isValid;
            for (const auto& ob : live->second.observers)
                if (kv.second.observers.find(ob.first) == kv.second.observers.end()) gone.emplace_back(ob.first, ob.second);
            for (const auto& ob : kv.second.observers) {
                const auto had = live->second.observers.find(ob.first);
                if (had == live->second.observers.end() || had->second != ob.second) come.emplace_back(ob.first, ob.second);
            }
        }
        for (const auto& ob : gone) map->RemoveObserver(ob.first, ob.second, kv.first);
        for (const auto& ob : come) map->AddObserver(ob.first, ob.second, kv.first);
        if (kv.second.isValid != wasValid || isNew) map->UpdateStatus(kv.first, kv.second.isValid);
    }
    for (const auto& kv : gv.nodes) {      // Node::SetPoint3d(id2d, id3d, false) for every keypoint that changed its point
        auto* node = graph->Get(kv.first);
        std::vector<std::pair<unsigned, unsigned>> changed;
        {
            const auto& live = node->GetPoints3d();
            for (const auto& p : kv.second.points3d) {
                const auto had = live.find(p.first);
                if (had == live.end() || had->second != p.second) changed.push_back(p);
            }
        }
        for (const auto& p : changed) node->SetPoint3d(p.first, p.second, false);
    }
#ifdef EACHAM_GLUE_TIMING
    {
        const auto t_d = std::chrono::steady_clock::now();
        tm.conv += std::chrono::duration<double, std::milli>(t_b - t_a).count();
        tm.call += std::chrono::duration<double, std::milli>(t_c - t_b).count();
        tm.back += std::chrono::duration<double, std::milli>(t_d - t_c).count();
        ++tm.n;
    }
#endif
    return rep;
}

}  // namespace glue
}  // namespace hip

#if defined(EACHAM_GLUE_REFERENCE_TYPES) || defined(EACHAM_GLUE_STANDINS)
// The reference's own signatures (BundleAdjuster.h:13-17, Triangulator.h:41-43): drop-in definitions.
inline void RefineBA(const int currentFrameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map, cv::Mat& K,
                     const OptimizerConfig& config) {
    (void)hip::glue::RefineBA(currentFrameId, graph, map, K, config);
}
inline void TriangulateFrame(const unsigned frameId, std::shared_ptr<graph_t> graph, std::shared_ptr<Map> map, const cv::Mat& K,
                             const unsigned minObservers, const float maxReprError, const float minTriAngle) {
    (void)hip::glue::TriangulateFrame(frameId, graph, map, K, minObservers, maxReprError, minTriAngle);
}
#endif

}  // namespace eacham
