// Test driver for include/eacham/SfmIO.hpp (header-only, no GPU): run as  io_driver <dir>
//   <dir>/config.json   -> <dir>/config.out      the parsed SfmConfig, one "name value" per line
//   <dir>/positions.txt -> <dir>/transform.json  (SavePositions) and <dir>/transforms_nerf.json (TransformToNerf)
//   <dir>/numbers.txt   -> <dir>/numbers.out     FormatDouble of every hex-float line
#include <eacham/SfmIO.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

using namespace eacham::hip::io;

static void put(std::ofstream& f, const char* name, double v) { f << name << " " << FormatDouble(v) << "\n"; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string dir = std::string(argv[1]) + "/";
    try {
        {
            const SfmConfig c = SfmConfig::Parse(LoadJson(dir + "config.json"));
            std::ofstream f(dir + "config.out");
            f << "imagesPath " << c.imagesPath << "\n" << "outputTransformPath " << c.outputTransformPath << "\n";
            f << "minFeaturesCount " << c.minFeaturesCount << "\n" << "maxFeaturesCount " << c.maxFeaturesCount << "\n";
            put(f, "inliersRatio", c.inliersRatio);
            f << "maxDataSize " << c.maxDataSize << "\n" << "initialMinInliers " << c.initialMinInliers << "\n";
            put(f, "initialMaxReprError", c.initialMaxReprError);
            put(f, "initialMinTriAngle", c.initialMinTriAngle);
            put(f, "maxReprError", c.maxReprError);
            put(f, "minTriAngle", c.minTriAngle);
            f << "minPnpInliers " << c.minPnpInliers << "\n";
            for (const auto* o : {&c.refineOpt, &c.globalOpt}) {
                f << "method " << o->method << "\n" << "maxIter " << o->maxIter << "\n";
                put(f, "maxTolerance", o->maxTolerance);
                put(f, "delta", o->delta);
                f << "usePreconditioner " << (o->usePreconditioner ? 1 : 0) << "\n";
            }
            f << "ui " << (c.ui ? 1 : 0) << "\n";
        }
        if (argc > 2 && std::string(argv[2]) == "config-only") return 0;
        {
            std::ifstream in(dir + "positions.txt");
            int n = 0;
            std::string tok;
            in >> n;
            float s[6];
            for (float& v : s) { in >> tok; v = std::strtof(tok.c_str(), nullptr); }
            std::map<unsigned, std::pair<std::string, Pose>> positions;
            for (int k = 0; k < n; ++k) {
                unsigned id;
                std::string path;
                in >> id >> path;
                Pose T;
                for (double& v : T) { in >> tok; v = std::strtod(tok.c_str(), nullptr); }
                positions[id] = {path, T};
            }
            SavePositions(dir + "transform.json", positions, s[0], s[1], s[2], s[3], s[4], s[5]);
            TransformToNerf(dir);
        }
        {
            std::ifstream in(dir + "numbers.txt");
            std::ofstream out(dir + "numbers.out");
            std::string tok;
            while (in >> tok) out << FormatDouble(std::strtod(tok.c_str(), nullptr)) << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "io_driver: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
