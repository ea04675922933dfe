// Test driver for include/eacham/TwoViewHip.hpp and PnPHip.hpp.
//   twoview_driver decompose   < "H(9) K(9)", "E(9)" or "R(9)" lines -> prints the decompositions / the Rodrigues vector (host-only math)
//   twoview_driver pnp <in.bin> <out.bin>                          -> SolvePnPRansac through the C-ABI (GPU)
//   twoview_driver pipeline <in.bin> <out.bin>                    -> FindEssentialMat / FindHomography / RecoverPose /
//                                                                    DecomposeHomographyMat through the C-ABI (GPU)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "eacham/PnPHip.hpp"
#include "eacham/TwoViewHip.hpp"

using namespace eacham::hip;

template <class T> static std::vector<T> rd(std::ifstream& f, size_t n) {
    std::vector<T> v(n);
    f.read((char*)v.data(), sizeof(T) * n);
    return v;
}
template <class T> static void wr(std::ofstream& f, const std::vector<T>& v) {
    int64_t n = (int64_t)v.size();
    f.write((char*)&n, sizeof(n));
    f.write((const char*)v.data(), sizeof(T) * v.size());
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "decompose")) {
        std::string kind;
        while (std::cin >> kind) {
            if (kind == "H") {
                Mat3 H;
                double K[9];
                for (double& x : H) std::cin >> x;
                for (double& x : K) std::cin >> x;
                const auto sols = DecomposeHomographyMat(H, K);
                std::printf("H %zu\n", sols.size());
                for (const auto& m : sols) {
                    for (double x : m.R) std::printf("%.17g ", x);
                    for (double x : m.t) std::printf("%.17g ", x);
                    for (double x : m.n) std::printf("%.17g ", x);
                    std::printf("\n");
                }
            } else if (kind == "R") {
                Mat3 R;
                for (double& x : R) std::cin >> x;
                const Vec3 r = RodriguesFromMatrix(R);
                std::printf("R %.17g %.17g %.17g\n", r[0], r[1], r[2]);
            } else {
                Mat3 E, R1, R2;
                Vec3 t;
                for (double& x : E) std::cin >> x;
                DecomposeEssentialMat(E, R1, R2, t);
                std::printf("E\n");
                for (double x : R1) std::printf("%.17g ", x);
                for (double x : R2) std::printf("%.17g ", x);
                for (double x : t) std::printf("%.17g ", x);
                std::printf("\n");
            }
        }
        return 0;
    }
    if (argc < 4) return 2;
    std::ifstream in(argv[2], std::ios::binary);
    std::ofstream out(argv[3], std::ios::binary);
    Context ctx(0);
    if (!strcmp(argv[1], "pnp")) {   // n, object (n x 3), image (n x 2), K9 -> ok, iterations, R, rvec, t, inliers
        int32_t n;
        in.read((char*)&n, sizeof(n));
        const auto obj = rd<double>(in, 3 * (size_t)n), img = rd<double>(in, 2 * (size_t)n), K9 = rd<double>(in, 9);
        const PnPResult r = SolvePnPRansac(ctx, obj, img, K9.data(), 10000, 4.0f, 0.999, 5);
        std::vector<double> pose{(double)r.ok, (double)r.iterations};
        pose.insert(pose.end(), r.R.begin(), r.R.end());
        pose.insert(pose.end(), r.rvec.begin(), r.rvec.end());
        pose.insert(pose.end(), r.t.begin(), r.t.end());
        wr(out, pose);
        wr(out, std::vector<int32_t>(r.inliers.begin(), r.inliers.end()));
        std::printf("twoview driver ok\n");
        return 0;
    }
    for (;;) {   // one scene per record (a general one: E branch, a planar one: H branch, ...)
        int32_t n;
        if (!in.read((char*)&n, sizeof(n))) break;
        const auto uv1 = rd<double>(in, 2 * (size_t)n), uv2 = rd<double>(in, 2 * (size_t)n), K9 = rd<double>(in, 9);
        const double K4[4] = {K9[0], K9[4], K9[2], K9[5]};
        const RobustModel Em = FindEssentialMat(ctx, uv1, uv2, K4, 1000, 7);
        const RobustModel Hm = FindHomography(ctx, uv1, uv2, 100, 7);
        std::vector<double> meta{(double)Em.ok, (double)Em.inliers, (double)Em.median, (double)Hm.ok, (double)Hm.inliers, (double)Hm.median,
                                 (double)Em.iterations, (double)Hm.iterations};
        wr(out, meta);
        wr(out, std::vector<double>(Em.model.begin(), Em.model.end()));
        wr(out, std::vector<double>(Hm.model.begin(), Hm.model.end()));
        wr(out, Em.mask);
        wr(out, Hm.mask);
        const RecoveredPose rp = RecoverPose(ctx, Em.model, uv1, uv2, K9.data(), 50.0, &Em.mask);
        std::vector<double> pose(rp.R.begin(), rp.R.end());
        pose.insert(pose.end(), rp.t.begin(), rp.t.end());
        pose.push_back(rp.good);
        wr(out, pose);
        // the homography branch of RecoverPoseTwoView (:92-150): decompose, triangulate every match under each solution, best count
        const auto sols = DecomposeHomographyMat(Hm.model, K9.data());
        std::vector<double> T;
        for (const auto& m : sols) {
            const double M[16] = {m.R[0], m.R[1], m.R[2], m.t[0], m.R[3], m.R[4], m.R[5], m.t[1], m.R[6], m.R[7], m.R[8], m.t[2], 0, 0, 0, 1};
            T.insert(T.end(), M, M + 16);
        }
        const auto tv = TwoViewPoints(ctx, uv1, uv2, K9.data(), T, 4.0f, 0.0174533f, true);
        const int best = BestTwoViewSolution(tv);
        std::vector<double> hb{(double)sols.size(), (double)best};
        for (const auto& s : tv) hb.push_back((double)s.matches.size());
        for (const auto& m : sols) {
            hb.insert(hb.end(), m.R.begin(), m.R.end());
            hb.insert(hb.end(), m.t.begin(), m.t.end());
        }
        wr(out, hb);
    }
    std::printf("twoview driver ok\n");
    return 0;
}
