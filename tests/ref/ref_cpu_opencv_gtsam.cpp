// ref_cpu_opencv_gtsam.cpp — the reference's two hot spots on ITS OWN libraries, fed with this repository's fixtures.
//
//   ref_cpu_opencv_gtsam match <in.bin> <out.bin> [flann|bf]   FeatureMatcherFlann::Match + the pair loop of apps/sfm/main.cpp:84-147
//   ref_cpu_opencv_gtsam ba    <in.bin> <out.bin>              the factor graph and optimiser of BundleAdjuster.cpp:47-216
//   ref_cpu_opencv_gtsam rng   <out.bin>                       cv::RNG((uint64)-1): 64 raw draws, 64 uniform(0, 100), 64 uniform(0, 2000)
//   ref_cpu_opencv_gtsam twoview <in.bin> <out.bin>            cv::findEssentialMat / findHomography (LMEDS) / recoverPose / decomposeHomographyMat
//                                                              exactly as ReconstructionManager.cpp:57-61, :75, :92, :154 call them
//   ref_cpu_opencv_gtsam pnp   <in.bin> <out.bin>              cv::solvePnPRansac(..., 10000, 4.0f, 0.999f, inliers, SOLVEPNP_EPNP), :227-228
// The last three pin what include/eacham/CvSampling.hpp (the sample stream), oracle/solve_oracle.c (the minimal solvers) and the
// LMedS / RANSAC loops of TwoViewHip.hpp / PnPHip.hpp restate from memory: run_ref.py holds the models, masks and inlier lists
// against the library's own estimators on the same seeded correspondences (OpenCV's getSubset is private to the registrators:
// what can be observed from outside is the raw generator and the estimators' results, which depend on every draw).
//
// NEVER COMPILED in this project's image (OpenCV 4.5.5 / GTSAM 4.1.1 are absent; tests/ref/CMakeLists.txt builds it only when
// find_package finds both) and therefore written from memory of those libraries' APIs — expect to fix a call or two. It exists
// because it is the only route by which "parity unpinned" can ever end: tests/ref/run_ref.py exports the committed golden inputs
// (tests/golden/*.npz), runs this binary and holds its output against the CPU oracle's (and so against the device path):
//   match  "flann" is what the reference runs (randomised KD-trees: approximate, not reproducible run to run),
//          "bf" is cv::BFMatcher(NORM_L2), the exact 2-NN the oracle restates: indices must agree exactly with the oracle's.
//   ba     GTSAM's own linearisation / elimination / lambda policy: iterations, errors, poses and points to 1e-5 relative.
// The arrays are the C-ABI's (include/eacham_hip.h: eacham_ba_problem, the CSR match graph); binary files are little-endian:
// a sequence of { int64 count, payload } records in the order the readers below take them.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#include <gtsam/geometry/Cal3_S2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/linear/PCGSolver.h>
#include <gtsam/linear/Preconditioner.h>
#include <gtsam/nonlinear/DoglegOptimizer.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/slam/GeneralSFMFactor.h>

template <class T>
static std::vector<T> rd(std::ifstream& f) {
    int64_t n = 0;
    f.read((char*)&n, sizeof(n));
    std::vector<T> v((size_t)n);
    f.read((char*)v.data(), sizeof(T) * (size_t)n);
    return v;
}
template <class T>
static void wr(std::ofstream& f, const std::vector<T>& v) {
    const int64_t n = (int64_t)v.size();
    f.write((const char*)&n, sizeof(n));
    f.write((const char*)v.data(), sizeof(T) * v.size());
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- matching: frames (n x dim float32 each), ordered list of unordered pairs, ratio / thresholds --------------------------
typedef std::unordered_map<unsigned, unsigned> match_t;

static match_t match_directed(const cv::Ptr<cv::DescriptorMatcher>& matcher, const cv::Mat& d1, const cv::Mat& d2, double ratio) {
    match_t out;                                               // FeatureMatcherFlann.cpp:14-30
    std::vector<std::vector<cv::DMatch>> knn;
    if (d2.rows < 2) return out;                               // (the reference would read m[1] out of bounds)
    matcher->knnMatch(d1, d2, knn, 2);
    for (const auto& m : knn)
        if (m.size() >= 2 && m[0].distance / m[1].distance < ratio) out.insert({(unsigned)m[0].queryIdx, (unsigned)m[0].trainIdx});
    return out;
}

static int run_match(const char* in, const char* out, const std::string& kind) {
    std::ifstream f(in, std::ios::binary);
    if (!f) return 2;
    const auto hdr = rd<int32_t>(f);                           // n_frames, dim, min_dir, min_mutual
    const auto ratio = rd<double>(f);
    const int F = hdr.at(0), dim = hdr.at(1), min_dir = hdr.at(2), min_mutual = hdr.at(3);
    std::vector<cv::Mat> frames;
    for (int k = 0; k < F; ++k) {
        const auto d = rd<float>(f);
        cv::Mat m((int)(d.size() / dim), dim, CV_32F);
        std::memcpy(m.data, d.data(), sizeof(float) * d.size());
        frames.push_back(m);
    }
    const auto pairs = rd<int32_t>(f);
    cv::Ptr<cv::DescriptorMatcher> matcher = kind == "bf" ? cv::DescriptorMatcher::create("BruteForce")   // NORM_L2
                                                          : cv::DescriptorMatcher::create("FlannBased");  // FeatureMatcherFlann.cpp:11
    std::vector<int32_t> counts, stats;
    std::vector<uint32_t> q, t;
    const double t0 = now_ms();
    for (size_t p = 0; p + 1 < pairs.size(); p += 2) {          // main.cpp:84-147 for the unordered pair (f1, f2)
        const cv::Mat &d1 = frames.at(pairs[p]), &d2 = frames.at(pairs[p + 1]);
        const match_t m12 = match_directed(matcher, d1, d2, ratio.at(0)), m21 = match_directed(matcher, d2, d1, ratio.at(0));
        std::map<unsigned, unsigned> mutual;                    // sorted by query index: the canonical form
        if ((int)m12.size() >= min_dir && (int)m21.size() >= min_dir)                                    // :111
            for (const auto& [a, b] : m12) {
                const auto back = m21.find(b);
                if (back != m21.end() && back->second == a) mutual[a] = b;                               // :133-140
            }
        const bool edge = (int)m12.size() >= min_dir && (int)m21.size() >= min_dir && (int)mutual.size() > min_mutual;   // :142
        // |mutual| for the statistics is counted whatever the direction thresholds say
        int nm = 0;
        for (const auto& [a, b] : m12) {
            const auto back = m21.find(b);
            nm += back != m21.end() && back->second == a;
        }
        counts.push_back(edge ? (int32_t)mutual.size() : 0);
        stats.insert(stats.end(), {(int32_t)m12.size(), (int32_t)m21.size(), (int32_t)nm, edge ? 1 : 0});
        if (edge)
            for (const auto& [a, b] : mutual) q.push_back(a), t.push_back(b);
    }
    const double ms = now_ms() - t0;
    std::ofstream o(out, std::ios::binary);
    wr(o, counts); wr(o, q); wr(o, t); wr(o, stats);
    wr(o, std::vector<double>{ms});
    std::printf("match (%s): %zu pairs in %.1f ms\n", kind.c_str(), pairs.size() / 2, ms);
    return 0;
}

// ---- bundle adjustment: the arrays of eacham_ba_problem + OptimizerConfig -----------------------------------------------------
static gtsam::noiseModel::Diagonal::shared_ptr noise6(float pos, float rotDeg) {                          // BundleAdjuster.cpp:28-33
    const float rot = rotDeg * 3.141592f / 180.0f;
    return gtsam::noiseModel::Diagonal::Sigmas((gtsam::Vector(6) << gtsam::Vector3::Constant(rot), gtsam::Vector3::Constant(pos)).finished());
}

static int run_ba(const char* in, const char* out) {
    std::ifstream f(in, std::ios::binary);
    if (!f) return 2;
    const auto T_wc = rd<double>(f);            // n_cams x 16 row-major world->camera
    const auto fixed = rd<int32_t>(f);
    const auto pts = rd<double>(f);             // n_points x 3
    const auto observers = rd<int32_t>(f);      // global observer count per point
    const auto obs_cam = rd<uint32_t>(f), obs_pt = rd<uint32_t>(f);
    const auto uv = rd<double>(f);              // n_obs x 2
    const auto K4 = rd<double>(f);              // fx fy cx cy
    const auto opt = rd<double>(f);             // method (0 LM, 1 DogLeg), maxIter, maxTolerance, delta, usePreconditioner
    const int nc = (int)fixed.size(), nl = (int)observers.size(), no = (int)obs_cam.size();
    gtsam::Cal3_S2 cal(K4.at(0), K4.at(1), 0.0, K4.at(2), K4.at(3));                                      // :47-49
    gtsam::NonlinearFactorGraph graph;
    gtsam::Values initial;
    const auto poseNoise = gtsam::noiseModel::Robust::Create(gtsam::noiseModel::mEstimator::Huber::Create(2.5f), noise6(0.35f, 45.0f));   // :60-63
    const auto fixedNoise = noise6(0.0001f, 0.0001f);                                                      // :71
    std::vector<char> seen(nl, 0);
    // The reference adds a frame's prior, then that frame's factors (and a landmark's prior when it is first seen); the
    // observations arrive grouped by camera in the fixtures (BaArrays.from_scene), so one pass in array order does the same.
    std::vector<char> cam_added(nc, 0);
    auto add_camera = [&](int c) {
        if (cam_added[c]) return;
        cam_added[c] = 1;
        Eigen::Matrix4d M;
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) M(r, k) = T_wc[16 * (size_t)c + 4 * r + k];
        const gtsam::Pose3 pose(M.inverse());                                                              // :65 camera->world
        initial.insert(gtsam::Symbol('x', c), pose);
        if (fixed[c]) graph.addPrior(gtsam::Symbol('x', c), pose, fixedNoise);                             // :69-77
        else graph.addPrior(gtsam::Symbol('x', c), pose, poseNoise);
    };
    for (int c = 0; c < nc; ++c) add_camera(c);   // (every camera of the problem carries a prior, observed or not)
    for (int o = 0; o < no; ++o) {
        const int c = (int)obs_cam[o], j = (int)obs_pt[o];
        const auto pix = gtsam::noiseModel::Robust::Create(gtsam::noiseModel::mEstimator::Huber::Create(3.0f),
                                                           gtsam::noiseModel::Isotropic::Sigma(2, 1.5f));  // :89-91
        graph.emplace_shared<gtsam::GeneralSFMFactor2<gtsam::Cal3_S2>>(gtsam::Point2(uv[2 * (size_t)o], uv[2 * (size_t)o + 1]), pix,
                                                                       gtsam::Symbol('x', c), gtsam::Symbol('l', j), gtsam::Symbol('K', 0));   // :95-98
        if (!seen[j]) {                                                                                    // :100-117
            seen[j] = 1;
            const gtsam::Point3 X(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]);
            initial.insert(gtsam::Symbol('l', j), X);
            const auto n = (size_t)(observers[j] > 0 ? observers[j] : 1);
            const auto prior = gtsam::noiseModel::Robust::Create(gtsam::noiseModel::mEstimator::Huber::Create(3.0f / n),
                                                                 gtsam::noiseModel::Isotropic::Sigma(3, 1.0f / n));
            graph.addPrior(gtsam::Symbol('l', j), X, prior);
        }
    }
    int used = 0;
    for (char s : seen) used += s;
    std::vector<double> meta;   // status (0 done, 1 skipped), iterations, initial error, final error, final lambda, ms
    std::ofstream o(out, std::ios::binary);
    if (used < 50) {                                                                                       // :166-169
        meta = {1, 0, 0, 0, 0, 0};
        wr(o, meta); wr(o, T_wc); wr(o, pts); wr(o, K4);
        return 0;
    }
    initial.insert(gtsam::Symbol('K', 0), cal);                                                            // :171
    graph.emplace_shared<gtsam::PriorFactor<gtsam::Cal3_S2>>(gtsam::Symbol('K', 0), cal,
        gtsam::noiseModel::Diagonal::Sigmas((gtsam::Vector(5) << 25, 25, 0.00001, 0.0001, 0.0001).finished()));   // :173-178
    const double t0 = now_ms();
    gtsam::Values result;
    double lambda = 0.0;
    size_t iterations = 0;
    if (opt.at(0) == 0.0) {                                                                                // :182-202
        gtsam::LevenbergMarquardtParams params;
        gtsam::LevenbergMarquardtParams::SetCeresDefaults(&params);
        params.absoluteErrorTol = opt.at(2);
        params.relativeErrorTol = opt.at(2);
        params.maxIterations = (int)opt.at(1);
        if (opt.at(4) != 0.0) {
            params.linearSolverType = gtsam::NonlinearOptimizerParams::Iterative;
            auto pcg = boost::make_shared<gtsam::PCGSolverParameters>();
            pcg->preconditioner_ = boost::make_shared<gtsam::BlockJacobiPreconditionerParameters>();
            pcg->setEpsilon_abs(1e-10);
            pcg->setEpsilon_rel(1e-10);
            params.iterativeParams = pcg;
        }
        gtsam::LevenbergMarquardtOptimizer lm(graph, initial, params);
        result = lm.optimize();                                                                            // :216
        lambda = lm.lambda();
        iterations = lm.iterations();
    } else {                                                                                               // :204-214
        gtsam::DoglegParams params;
        params.absoluteErrorTol = opt.at(2);
        params.relativeErrorTol = opt.at(2);
        params.maxIterations = (int)opt.at(1);
        params.setDeltaInitial(opt.at(3));
        gtsam::DoglegOptimizer dl(graph, initial, params);
        result = dl.optimize();
        lambda = dl.getDelta();
        iterations = dl.iterations();
    }
    const double ms = now_ms() - t0;
    meta = {0, (double)iterations, graph.error(initial), graph.error(result), lambda, ms};                 // :218-219
    std::vector<double> outT(T_wc.size()), outP(pts), outK(4);
    const gtsam::Cal3_S2 k = result.at<gtsam::Cal3_S2>(gtsam::Symbol('K', 0));                             // :221-227
    outK = {k.fx(), k.fy(), k.px(), k.py()};
    for (int j = 0; j < nl; ++j)
        if (seen[j]) {                                                                                     // :229-236
            const gtsam::Point3 X = result.at<gtsam::Point3>(gtsam::Symbol('l', j));
            outP[3 * (size_t)j] = X.x(), outP[3 * (size_t)j + 1] = X.y(), outP[3 * (size_t)j + 2] = X.z();
        }
    for (int c = 0; c < nc; ++c) {                                                                         // :238-248
        const Eigen::Matrix4d M = result.at<gtsam::Pose3>(gtsam::Symbol('x', c)).matrix().inverse();
        for (int r = 0; r < 4; ++r)
            for (int q = 0; q < 4; ++q) outT[16 * (size_t)c + 4 * r + q] = M(r, q);
    }
    wr(o, meta); wr(o, outT); wr(o, outP); wr(o, outK);
    std::printf("ba: %zu iterations, error %.6g -> %.6g, %.1f ms\n", iterations, meta[2], meta[3], ms);
    return 0;
}

// ---- the estimators' random stream and the estimators themselves ------------------------------------------------------------
static int run_rng(const char* out) {
    std::ofstream o(out, std::ios::binary);
    cv::RNG a((uint64_t)-1), b((uint64_t)-1), c((uint64_t)-1);   // what LMeDSPointSetRegistrator::run / RANSACPointSetRegistrator::run seed
    std::vector<uint32_t> raw(64);
    std::vector<int32_t> u100(64), u2000(64);
    for (int i = 0; i < 64; ++i) raw[i] = a.next(), u100[i] = b.uniform(0, 100), u2000[i] = c.uniform(0, 2000);
    wr(o, raw); wr(o, u100); wr(o, u2000);
    return 0;
}

static std::vector<double> mat_to_vec(const cv::Mat& m) {
    cv::Mat d;
    m.convertTo(d, CV_64F);
    return std::vector<double>((const double*)d.datastart, (const double*)d.dataend);
}

static int run_twoview(const char* in, const char* out) {
    std::ifstream f(in, std::ios::binary);
    if (!f) return 2;
    const auto Kv = rd<double>(f);                              // 3 x 3 row-major
    const auto a = rd<float>(f), b = rd<float>(f);             // n x 2 each: cv::Point2f, as GetMatchedPoints hands them over
    const int n = (int)a.size() / 2;
    std::vector<cv::Point2f> pts1(n), pts2(n);
    for (int i = 0; i < n; ++i) pts1[i] = {a[2 * i], a[2 * i + 1]}, pts2[i] = {b[2 * i], b[2 * i + 1]};
    const cv::Mat K = (cv::Mat_<double>(3, 3) << Kv[0], Kv[1], Kv[2], Kv[3], Kv[4], Kv[5], Kv[6], Kv[7], Kv[8]);
    const double t0 = now_ms();
    cv::Mat mask, mask2;
    const cv::Mat E = cv::findEssentialMat(pts1, pts2, K.at<double>(0, 0), cv::Point2d{K.at<double>(0, 2), K.at<double>(1, 2)},
                                           cv::LMEDS, 0.99f, 4.0f, 1000, mask);                              // :57-61
    const cv::Mat H = cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999);                    // :75
    cv::Mat R, t, maskPose;
    const int nPose = E.empty() ? 0 : cv::recoverPose(E, pts1, pts2, K, R, t, maskPose);                   // :154
    std::vector<cv::Mat> Rs, ts, ns;
    if (!H.empty()) cv::decomposeHomographyMat(H, K, Rs, ts, ns);                                          // :92
    const double ms = now_ms() - t0;
    std::ofstream o(out, std::ios::binary);
    wr(o, mat_to_vec(E)); wr(o, mat_to_vec(mask)); wr(o, mat_to_vec(H)); wr(o, mat_to_vec(mask2));
    wr(o, mat_to_vec(R)); wr(o, mat_to_vec(t)); wr(o, std::vector<double>{(double)nPose, (double)Rs.size(), ms});
    std::vector<double> dec;
    for (size_t k = 0; k < Rs.size(); ++k) {
        for (double v : mat_to_vec(Rs[k])) dec.push_back(v);
        for (double v : mat_to_vec(ts[k])) dec.push_back(v);
        for (double v : mat_to_vec(ns[k])) dec.push_back(v);
    }
    wr(o, dec);
    std::printf("twoview: %d matches, E inliers %d, H inliers %d, recoverPose %d, %zu homography solutions, %.1f ms\n", n,
                cv::countNonZero(mask), cv::countNonZero(mask2), nPose, Rs.size(), ms);
    return 0;
}

static int run_pnp(const char* in, const char* out) {
    std::ifstream f(in, std::ios::binary);
    if (!f) return 2;
    const auto Kv = rd<double>(f);
    const auto obj = rd<float>(f), img = rd<float>(f);         // n x 3 cv::Point3f, n x 2 cv::Point2f
    const int n = (int)img.size() / 2;
    std::vector<cv::Point3f> pts3d(n);
    std::vector<cv::Point2f> pts2d(n);
    for (int i = 0; i < n; ++i) pts3d[i] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]}, pts2d[i] = {img[2 * i], img[2 * i + 1]};
    const cv::Mat K = (cv::Mat_<double>(3, 3) << Kv[0], Kv[1], Kv[2], Kv[3], Kv[4], Kv[5], Kv[6], Kv[7], Kv[8]);
    std::vector<double> distCoeffs = {0, 0, 0, 0};                                                        // :219
    cv::Mat rvec = cv::Mat_<double>(3, 1), t = cv::Mat_<double>(3, 1);
    std::vector<int> inliers;
    const double t0 = now_ms();
    const bool ok = cv::solvePnPRansac(pts3d, pts2d, K, distCoeffs, rvec, t, false, 10000, 4.0f, 0.999f, inliers, cv::SOLVEPNP_EPNP);  // :227-228
    const double ms = now_ms() - t0;
    std::ofstream o(out, std::ios::binary);
    wr(o, mat_to_vec(rvec)); wr(o, mat_to_vec(t));
    wr(o, std::vector<int32_t>(inliers.begin(), inliers.end()));
    wr(o, std::vector<double>{ok ? 1.0 : 0.0, ms});
    std::printf("pnp: %d correspondences, %zu inliers, ok %d, %.1f ms\n", n, inliers.size(), (int)ok, ms);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "match")) return run_match(argv[2], argv[3], argc > 4 ? argv[4] : "flann");
    if (argc >= 4 && !std::strcmp(argv[1], "ba")) return run_ba(argv[2], argv[3]);
    if (argc >= 3 && !std::strcmp(argv[1], "rng")) return run_rng(argv[2]);
    if (argc >= 4 && !std::strcmp(argv[1], "twoview")) return run_twoview(argv[2], argv[3]);
    if (argc >= 4 && !std::strcmp(argv[1], "pnp")) return run_pnp(argv[2], argv[3]);
    std::fprintf(stderr, "usage: %s match|ba|twoview|pnp <in.bin> <out.bin> [flann|bf]  |  rng <out.bin>\n", argv[0]);
    return 2;
}
