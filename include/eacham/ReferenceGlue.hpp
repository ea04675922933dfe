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

#include <algorithm>
#include <map>
#include <memory>
#ifdef EACHAM_GLUE_TIMING
#include <chrono>
#include <cstdio>
#endif
#include <set>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>
#include <unordered_map>
#include <string>

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

// RefineBA with the reference's argument list (BundleAdjuster.cpp:40-250), directly on the reference's objects: the graph walk
// of :57-162 — the window (the current frame and its valid factor neighbours, or every valid node for currentFrameId = -1), the
// landmark filter `status && observers >= 2` (:84), first-seen landmark registration (:100-117) — fills the plain arrays of
// eacham_ba_problem straight from Graph / Node / Map (no intermediate views, no copies to diff against), eacham_ba_solve runs,
// and :221-249 is replayed: K, UpdatePoint + UpdateStatus(true) of every landmark of the problem, SetTransform of every camera.
// A node's keypoint -> landmark map is walked in ascending keypoint order and the window's neighbours in ascending id (the
// reference iterates unordered_maps there: no order to keep; a fixed one makes the call reproducible).
template <class GraphT, class MapT, class MatT, class ConfigT>
inline RefineBAReport RefineBA(const int currentFrameId, const std::shared_ptr<GraphT>& graph, const std::shared_ptr<MapT>& map,
                               MatT& K, const ConfigT& config) {
#ifdef EACHAM_GLUE_TIMING
    struct Tm { double walk = 0, solve = 0, back = 0; long n = 0; ~Tm() { if (n) std::fprintf(stderr, "RefineBA glue: %ld calls, graph walk %.3f ms, eacham_ba_solve %.3f ms, write-back %.3f ms per call\n", n, walk / n, solve / n, back / n); } };
    static Tm tm;
    const auto t_a = std::chrono::steady_clock::now();
#endif
    std::vector<unsigned> frameIds;
    std::vector<double> camT, points, uv;
    std::vector<int32_t> camFixed, pointObservers;
    std::vector<uint32_t> obsCam, obsPoint;
    std::vector<unsigned> mapIds;                      // dense point index -> landmark id
    const auto& all = map->GetAll();                   // (Map::Get / GetStatus / GetObservers of the reference each take a lock and the last one copies the map)
    // the window's nodes first: their sizes bound the arrays and the landmark table
    using NodeP = decltype(graph->Get(0u));
    std::vector<std::pair<unsigned, NodeP>> window;
    if (currentFrameId > -1) {                          // local window (:123-145)
        auto* start = graph->Get((unsigned)currentFrameId);
        if (!start) throw std::runtime_error("Node is null");
        window.emplace_back((unsigned)currentFrameId, start);
        std::vector<unsigned> nb;
        for (const auto& f : start->GetFactors()) nb.push_back(f.first);
        std::sort(nb.begin(), nb.end());
        for (unsigned id : nb) {
            auto* node = graph->Get(id);
            if (!node) throw std::runtime_error("Node is null");
            if (node->IsValid()) window.emplace_back(id, node);
        }
    } else {                                            // global (:146-162)
        for (const auto& entry : graph->GetNodes())
            if (entry.second->IsValid()) window.emplace_back(entry.first, entry.second);
    }
    size_t max_obs = 0;
    for (const auto& w : window) max_obs += w.second->GetPoints3d().size();
    // landmark id -> dense point index, or SKIP for a landmark the filter of :84 drops (decided once per landmark: nothing
    // the walk reads changes while it runs). One table per thread, reused from call to call.
    constexpr uint32_t SKIP = 0xffffffffu;
    static thread_local IdTable mapIndex;
    mapIndex.reset(max_obs);
    camT.reserve(16 * window.size()), camFixed.reserve(window.size()), frameIds.reserve(window.size());
    obsCam.reserve(max_obs), obsPoint.reserve(max_obs), uv.reserve(2 * max_obs);
    constexpr unsigned NONE = 0xffffffffu;
    std::vector<unsigned> byKeypoint;                  // a node's keypoint -> landmark map laid out by keypoint index
    std::vector<std::pair<unsigned, unsigned>> beyond; // (entries whose keypoint index is past the node's features: an error if they survive the filter)
    auto observe = [&](uint32_t cam, unsigned id2d, unsigned id3d, size_t n_kps, const auto& kps) {
        bool fresh = false;
        uint32_t& where = mapIndex.slot(id3d, fresh);
        if (fresh) {
            const auto it = all.find(id3d);
            if (it == all.end()) throw std::runtime_error("Map: point is not found");
            if (!it->second.isValid || it->second.observers.size() < 2) {               // :84
                where = SKIP;
            } else {                                                                     // :100-117
                where = (uint32_t)mapIds.size();
                mapIds.push_back(id3d);
                points.push_back(it->second.point3d(0));
                points.push_back(it->second.point3d(1));
                points.push_back(it->second.point3d(2));
                pointObservers.push_back((int32_t)it->second.observers.size());
            }
        }
        if (where == SKIP) return;
        if ((size_t)id2d >= n_kps) throw std::runtime_error("RefineBA: keypoint out of range");
        obsCam.push_back(cam);
        obsPoint.push_back(where);
        uv.push_back((double)kps[id2d].x);
        uv.push_back((double)kps[id2d].y);
    };
    for (const auto& w : window) {                     // BundleAdjuster.cpp:57-121
        auto* node = w.second;
        const uint32_t cam = (uint32_t)frameIds.size();
        frameIds.push_back(w.first);
        double T[16];
        matrix_to_rows(node->GetTransform(), T);
        camT.insert(camT.end(), T, T + 16);
        camFixed.push_back(graph->IsFixed(w.first) ? 1 : 0);
        const auto& kps = node->GetFeatures();
        const size_t n_kps = kps.size();
        // ascending keypoint order without a sort: the map's entries dropped into a table indexed by keypoint
        byKeypoint.assign(n_kps, NONE);
        beyond.clear();
        for (const auto& kv : node->GetPoints3d()) {
            if ((size_t)kv.first < n_kps) byKeypoint[kv.first] = kv.second;
            else beyond.emplace_back(kv.first, kv.second);
        }
        for (size_t k = 0; k < n_kps; ++k)
            if (byKeypoint[k] != NONE) observe(cam, (unsigned)k, byKeypoint[k], n_kps, kps);
        std::sort(beyond.begin(), beyond.end());
        for (const auto& kv : beyond) observe(cam, kv.first, kv.second, n_kps, kps);
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
    prob.K[0] = K.template at<double>(0, 0); prob.K[1] = K.template at<double>(1, 1);       // :47-49
    prob.K[2] = K.template at<double>(0, 2); prob.K[3] = K.template at<double>(1, 2);
    eacham_ba_options opt{};
    const std::string method = config.method;
    if (method == "LM") opt.method = EACHAM_BA_LM;
    else if (method == "DogLeg") opt.method = EACHAM_BA_DOGLEG;
    else throw std::runtime_error("RefineBA: unknown method " + method);                   // the reference would dereference a null optimizer
    opt.max_iter = config.maxIter;
    opt.max_tolerance = config.maxTolerance;
    opt.delta = config.delta;
    opt.use_preconditioner = config.usePreconditioner ? 1 : 0;
    opt.min_landmarks = 50;                                                                 // :166
    std::vector<double> outT(camT.size() + 1), outP(points.size() + 1);
    eacham_ba_result res{};
    res.cam_T_wc = outT.data();
    res.points = outP.data();
    eacham_ctx* ctx = shared_context().get();
#ifdef EACHAM_GLUE_TIMING
    const auto t_b = std::chrono::steady_clock::now();
#endif
    const int rc = eacham_ba_solve(ctx, &prob, &opt, &res);
#ifdef EACHAM_GLUE_TIMING
    const auto t_c = std::chrono::steady_clock::now();
#endif
    if (rc != EACHAM_OK) throw std::runtime_error(std::string("eacham_hip: ") + eacham_last_error(ctx));
    rep.skipped = res.status == EACHAM_BA_SKIPPED;
    if (rep.skipped) return rep;  // fewer than 50 landmarks: the reference returns without touching anything (:166-169)
    rep.initial_error = res.initial_error;
    rep.final_error = res.final_error;
    rep.outer_iterations = res.outer_iterations;
    rep.inner_iterations = res.inner_iterations;
    K.template at<double>(0, 0) = res.K[0];   // :224-227
    K.template at<double>(1, 1) = res.K[1];
    K.template at<double>(0, 2) = res.K[2];
    K.template at<double>(1, 2) = res.K[3];
    using Vec3 = std::decay_t<decltype(map->Get(0u))>;
    for (size_t j = 0; j < mapIds.size(); ++j) {   // :229-241
        Vec3 X;
        X(0) = outP[3 * j], X(1) = outP[3 * j + 1], X(2) = outP[3 * j + 2];
        map->UpdatePoint(mapIds[j], X);
        map->UpdateStatus(mapIds[j], true);
    }
    for (size_t i = 0; i < frameIds.size(); ++i) {  // :243-248
        auto* node = graph->Get(frameIds[i]);
        using Mat4 = std::decay_t<decltype(node->GetTransform())>;
        Mat4 M;
        for (int r = 0; r < 4; ++r)
            for (int col = 0; col < 4; ++col) M(r, col) = outT[16 * i + 4 * r + col];
        node->SetTransform(M);
    }
#ifdef EACHAM_GLUE_TIMING
    {
        const auto t_d = std::chrono::steady_clock::now();
        tm.walk += std::chrono::duration<double, std::milli>(t_b - t_a).count();
        tm.solve += std::chrono::duration<double, std::milli>(t_c - t_b).count();
        tm.back += std::chrono::duration<double, std::milli>(t_d - t_c).count();
        ++tm.n;
    }
#endif
    return rep;
}

// TriangulateFrame with the reference's argument list (Triangulator.cpp:188-300), DIRECTLY on the reference's objects: the walk
// of :204-296 is restated on the live Graph / Node / Map — same reads, same mutations in the same order (SetPoint3d /
// AddObserver in the re-observation gate, Map::Add / RemoveObserver / UpdateStatus / SetPoint3d / AddObserver per accepted
// track) — and only the arithmetic leaves for the device, in two batches: every candidate's reprojection error before the
// gate is walked (CalcReprojectionError does not depend on what the gate mutates), every candidate track before the
// bookkeeping. Until round 4 this function converted the frame, its ~20 factor neighbours and every map point they reference
// into views (one small vector per map point), ran the view-based adapter of TriangulatorHip.hpp and diffed the views back:
// 2 ms per call around tens of microseconds of kernels, 4.0 of the 7.4 ms the incremental loop spent per frame.
// Neighbours are visited in ascending id and matches in ascending (m1, m2), as the adapter does (the reference iterates
// unordered_maps: no order to keep). `color`: the reference draws one per call from cv::RNG (:192-199, display only).
template <class GraphT, class MapT, class MatT>
inline TriangulateFrameReport TriangulateFrame(const unsigned frameId, const std::shared_ptr<GraphT>& graph,
                                               const std::shared_ptr<MapT>& map, const MatT& K, const unsigned minObservers,
                                               const float maxReprError, const float minTriAngle) {
#ifdef EACHAM_GLUE_TIMING
    struct Tm { double conv = 0, call = 0, back = 0, g_nbs = 0, g_p1 = 0, g_dev = 0; long n = 0, cands = 0, matches = 0; ~Tm() { std::fprintf(stderr, "TriangulateFrame glue: %ld calls, gate %.3f ms (neighbours %.3f, pass one %.3f, reprojection call %.3f; %.0f matches, %.0f candidates per call), tracks %.3f ms, bookkeeping %.3f ms per call\n", n, conv / n, g_nbs / n, g_p1 / n, g_dev / n, (double)matches / n, (double)cands / n, call / n, back / n); } };
    static Tm tm;
    const auto t_a = std::chrono::steady_clock::now();
#endif
    Context& ctx = shared_context();
    auto* current = graph->Get(frameId);
    if (!current) throw std::runtime_error("Node is null");
    using NodeT = std::remove_pointer_t<decltype(current)>;
    const double K4[4] = {K.template at<double>(0, 0), K.template at<double>(1, 1), K.template at<double>(0, 2), K.template at<double>(1, 2)};
    TriangulateFrameReport rep;
    auto keypoint = [](NodeT* n, unsigned k, double* out) {
        const auto& f = n->GetFeatures();
        if ((size_t)k >= f.size()) throw std::runtime_error("TriangulateFrame: keypoint out of range");
        out[0] = f[k].x;
        out[1] = f[k].y;
    };
    struct Nb {
        unsigned id;
        NodeT* node;
        std::vector<std::pair<unsigned, unsigned>> matches;
    };
    // (the neighbour records and the two keypoint-indexed tables belong to the thread and keep their storage from call to call:
    // ten neighbours' match lists are ten allocations per call otherwise; a table entry is current when its stamp is the neighbour's)
    static thread_local std::vector<Nb> nbsStore;
    static thread_local std::vector<unsigned> partnerOf, stampOf;
    static thread_local unsigned stamp = 0;
    const size_t n_cur = current->GetFeatures().size();
    if (partnerOf.size() < n_cur) partnerOf.resize(n_cur), stampOf.resize(n_cur, 0u);
    size_t n_nbs = 0;
    for (const auto& f : current->GetFactors()) {
        auto* other = graph->Get(f.first);
        if (!other) throw std::runtime_error("Node is null");
        if (!other->IsValid()) continue;                 // :208-211
        if (nbsStore.size() <= n_nbs) nbsStore.emplace_back();
        Nb& nb = nbsStore[n_nbs++];
        nb.id = f.first, nb.node = other;
        nb.matches.clear();
        nb.matches.reserve(f.second.matches.size());
        if (++stamp == 0) {                              // (the stamp wrapped: no entry is current)
            std::fill(stampOf.begin(), stampOf.end(), 0u);
            stamp = 1;
        }
        bool in_range = true;
        unsigned lo = 0xffffffffu, hi = 0;               // a factor's matches laid out by the current frame's keypoint: ascending order without a sort
        for (const auto& mm : f.second.matches) {
            if ((size_t)mm.first < n_cur) {
                partnerOf[mm.first] = mm.second, stampOf[mm.first] = stamp;
                lo = std::min(lo, mm.first), hi = std::max(hi, mm.first);
            } else {
                in_range = false;
            }
        }
        if (in_range) {
            for (size_t k = lo; k <= hi && lo != 0xffffffffu; ++k)
                if (stampOf[k] == stamp) nb.matches.emplace_back((unsigned)k, partnerOf[k]);
        } else {                                         // (a match past the frame's keypoints: kept in order; it is an error further down if it is used)
            for (const auto& mm : f.second.matches) nb.matches.emplace_back(mm.first, mm.second);
            std::sort(nb.matches.begin(), nb.matches.end());
        }
    }
    // ascending neighbour id: an index list over the thread's records (the records themselves stay where their storage is)
    std::vector<const Nb*> nbs(n_nbs);
    for (size_t i = 0; i < n_nbs; ++i) nbs[i] = &nbsStore[i];
    std::sort(nbs.begin(), nbs.end(), [](const Nb* a, const Nb* b) { return a->id < b->id; });
#ifdef EACHAM_GLUE_TIMING
    const auto t_g1 = std::chrono::steady_clock::now();
#endif
    // ---- the re-observation gate (:213-240): errors of all candidates in one call, then the walk ----
    // Pass one looks every match up ONCE (the partner's keypoint -> landmark map, the landmark in the map) and keeps what it found:
    // pass two used to repeat both hash lookups per match. A landmark's record is held by address: the gate adds observers to
    // records, never records to the map, so the addresses stay put and `observers.size()` read through them is the LIVE count.
    const auto& all = map->GetAll();
    using PointRec = std::remove_reference_t<decltype(all.begin()->second)>;
    std::vector<unsigned> cand3d;     // the partner's map point per candidate, walk order
    std::vector<const PointRec*> candRec;
    std::vector<uint32_t> cframe;
    std::vector<double> cpts, cuv;
    std::vector<char> isCand;         // per match, walk order
    size_t n_matches = 0;
    unsigned max_kp = 0;
    for (const Nb* nbp : nbs) {
        const Nb& nb = *nbp;
        n_matches += nb.matches.size();
        if (!nb.matches.empty()) max_kp = std::max(max_kp, nb.matches.back().first);   // (sorted by the current frame's keypoint)
    }
    isCand.reserve(n_matches);
    for (const Nb* nbp : nbs) {
        const Nb& nb = *nbp;
        const auto& p3 = nb.node->GetPoints3d();
        for (const auto& mm : nb.matches) {
            const auto has = p3.find(mm.second);
            isCand.push_back(has != p3.end());
            if (has == p3.end()) continue;
            const auto mp = all.find(has->second);
            if (mp == all.end()) throw std::runtime_error("Map: point is not found");
            cand3d.push_back(has->second);
            candRec.push_back(&mp->second);
            cframe.push_back(0);
            cpts.push_back(mp->second.point3d(0));
            cpts.push_back(mp->second.point3d(1));
            cpts.push_back(mp->second.point3d(2));
            double uv[2];
            keypoint(current, mm.first, uv);
            cuv.push_back(uv[0]);
            cuv.push_back(uv[1]);
        }
    }
    std::vector<float> cerr(cand3d.size() + 1);
    double Tcur[16];
    matrix_to_rows(current->GetTransform(), Tcur);
#ifdef EACHAM_GLUE_TIMING
    const auto t_g2 = std::chrono::steady_clock::now();
#endif
    ctx.check(eacham_reprojection_errors(ctx.get(), Tcur, 1, (int)cand3d.size(), cframe.data(), cpts.data(), cuv.data(), K4, cerr.data()));
#ifdef EACHAM_GLUE_TIMING
    {
        const auto t_g3 = std::chrono::steady_clock::now();
        tm.g_nbs += std::chrono::duration<double, std::milli>(t_g1 - t_a).count();
        tm.g_p1 += std::chrono::duration<double, std::milli>(t_g2 - t_g1).count();
        tm.g_dev += std::chrono::duration<double, std::milli>(t_g3 - t_g2).count();
        tm.cands += (long)cand3d.size();
        tm.matches += (long)n_matches;
    }
#endif
    // the observers of every keypoint of the current frame that goes on to triangulation, by keypoint index (ascending = the order
    // of the std::map this used to be); the table is the thread's own and only the entries a call touched are cleared again
    static thread_local std::vector<FlatMap> observersFull;
    std::vector<unsigned> touched;
    if (observersFull.size() <= (size_t)max_kp) observersFull.resize((size_t)max_kp + 1);
    struct ClearTouched {   // (also on the exceptional exits)
        std::vector<FlatMap>& table;
        std::vector<unsigned>& keys;
        ~ClearTouched() { for (unsigned k : keys) table[k].clear(); }
    } clearTouched{observersFull, touched};
    {
        size_t ci = 0, mi = 0;
        for (const Nb* nbp : nbs) {
        const Nb& nb = *nbp;
            for (const auto& mm : nb.matches) {
                if (isCand[mi++]) {   // (the gate only writes the CURRENT frame's points3d: a neighbour's is what pass one saw)
                    const unsigned id3d = cand3d[ci];
                    const PointRec* rec = candRec[ci];
                    const float err = cerr[ci++];
                    if (rec->observers.size() > 2 && err < maxReprError) {  // the LIVE observer count, in walk order (:218)
                        current->SetPoint3d(mm.first, id3d, false);
                        map->AddObserver(frameId, mm.first, id3d);
                        ++rep.reobserved;
                        continue;
                    }
                }
                FlatMap& obs = observersFull[mm.first];
                if (obs.empty()) touched.push_back(mm.first);
                obs[frameId] = mm.first;
                obs[nb.id] = mm.second;
            }
        }
    }
    std::sort(touched.begin(), touched.end());
#ifdef EACHAM_GLUE_TIMING
    const auto t_b = std::chrono::steady_clock::now();
#endif
    // ---- candidate tracks (:248-262): one call ----
    std::map<unsigned, uint32_t> frameRow;   // node id -> row of the transform table
    std::map<unsigned, NodeT*> nodeOf;
    std::vector<double> transforms, uv;
    std::vector<int32_t> trackPtr{0};
    std::vector<uint32_t> obsFrame;
    std::vector<const FlatMap*> trackObs;
    for (const unsigned kp : touched) {
        const FlatMap& observers = observersFull[kp];
        if (observers.size() < minObservers) continue;
        for (const auto& ob : observers) {
            auto ins = frameRow.insert({ob.first, (uint32_t)frameRow.size()});
            if (ins.second) {
                auto* n = graph->Get(ob.first);
                if (!n) throw std::runtime_error("Node is null");
                nodeOf[ob.first] = n;
                double T[16];
                matrix_to_rows(n->GetTransform(), T);
                transforms.insert(transforms.end(), T, T + 16);
            }
            obsFrame.push_back(ins.first->second);
            double p[2];
            keypoint(nodeOf[ob.first], ob.second, p);
            uv.push_back(p[0]);
            uv.push_back(p[1]);
        }
        trackPtr.push_back((int32_t)obsFrame.size());
        trackObs.push_back(&observers);
    }
    const int nTracks = (int)trackObs.size();
    std::vector<double> pts((size_t)nTracks * 3 + 3);
    std::vector<int32_t> status(nTracks + 1);
    std::vector<uint8_t> masks(obsFrame.size() + 1);
    ctx.check(eacham_triangulate_tracks(ctx.get(), transforms.data(), (int)frameRow.size(), nTracks, trackPtr.data(), obsFrame.data(),
                                        uv.data(), K4, maxReprError, minTriAngle, pts.data(), status.data(), masks.data()));
#ifdef EACHAM_GLUE_TIMING
    const auto t_c = std::chrono::steady_clock::now();
#endif
    // ---- map bookkeeping for accepted tracks (:270-296), on the live objects ----
    using Vec3 = std::decay_t<decltype(map->Get(0u))>;
    for (int t = 0; t < nTracks; ++t) {
        if (status[t] == 3) {
            Vec3 X;
            X(0) = pts[(size_t)t * 3], X(1) = pts[(size_t)t * 3 + 1], X(2) = pts[(size_t)t * 3 + 2];
            const unsigned mapPointId = map->Add(X);
            for (const auto& ob : *trackObs[t]) {
                NodeT* n = nodeOf[ob.first];
                const auto& p3 = n->GetPoints3d();
                const auto old = p3.find(ob.second);
                if (old != p3.end()) {
                    const unsigned oldId = old->second;
                    map->RemoveObserver(ob.first, ob.second, oldId);
                    map->UpdateStatus(oldId, false);
                }
                n->SetPoint3d(ob.second, mapPointId, false);
                map->AddObserver(ob.first, ob.second, mapPointId);
            }
            map->UpdateStatus(mapPointId, true);
            ++rep.added;
        }
        ++rep.total;
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
