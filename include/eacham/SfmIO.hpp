// SfmIO.hpp — host-side IO around the hot path, header-only, no dependencies:
// the reference's config schema and its transform.json / transforms_nerf.json writers.
//
//   SfmConfig::Parse     modules/sfm/config/SfmConfig.h:27-71 (its quirks kept: see the comments there)
//   SavePositions        modules/sfm/utils/Saver.h:13-73, called at apps/sfm/main.cpp:259-264
//   TransformToNerf      apps/sfm/TransformToNerf.cpp:40-66
//
// The reference reads and writes JSON with nlohmann::json, an un-vendored Conan dependency that is not in this
// tree. Its output format (`file << std::setw(4) << j`: keys in lexicographic order, 4-space indent, one array
// element per line, doubles as the shortest string that round-trips, with ".0" when that string looks like an
// integer) is restated here from its documented behaviour: parity unpinned. Python mirror: eacham_amd/sfm_io.py
// (tests/test_sfm_io.py holds the two against each other byte for byte).
#pragma once

#include <array>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace eacham {
namespace hip {
namespace io {

// ---- a JSON value just large enough for the two documents ---------------------------------------------
struct Json {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    long long i = 0;
    double f = 0.0;
    std::string s;
    std::vector<Json> a;
    std::map<std::string, Json> o;  // lexicographic, like nlohmann's default object type

    Json() = default;
    Json(bool v) : kind(Bool), b(v) {}
    Json(int v) : kind(Int), i(v) {}
    Json(long long v) : kind(Int), i(v) {}
    Json(double v) : kind(Float), f(v) {}
    Json(const char* v) : kind(String), s(v) {}
    Json(const std::string& v) : kind(String), s(v) {}
    static Json array() { Json j; j.kind = Array; return j; }
    static Json object() { Json j; j.kind = Object; return j; }

    Json& operator[](const std::string& key) {
        if (kind == Null) kind = Object;
        if (kind != Object) throw std::runtime_error("json: not an object");
        return o[key];
    }
    const Json& at(const std::string& key) const {
        if (kind != Object) throw std::runtime_error("json: not an object");
        const auto it = o.find(key);
        if (it == o.end()) throw std::runtime_error("json: key '" + key + "' not found");
        return it->second;
    }
    void push_back(Json v) {
        if (kind == Null) kind = Array;
        if (kind != Array) throw std::runtime_error("json: not an array");
        a.push_back(std::move(v));
    }
    double number() const {
        if (kind == Int) return (double)i;
        if (kind == Float) return f;
        throw std::runtime_error("json: not a number");
    }
    long long integer() const {
        if (kind == Int) return i;
        if (kind == Float) return (long long)f;
        throw std::runtime_error("json: not a number");
    }
    const std::string& string() const {
        if (kind != String) throw std::runtime_error("json: not a string");
        return s;
    }
    bool boolean() const {
        if (kind != Bool) throw std::runtime_error("json: not a boolean");
        return b;
    }
};

inline std::string FormatDouble(double x) {  // shortest round-trip digits, ".0" for integer-looking values
    if (!std::isfinite(x)) return "null";
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof(buf), x);  // shortest round-trip (C++17)
    std::string out(buf, r.ptr);
    // std::to_chars picks fixed or scientific by length; nlohmann (and Python's repr) print fixed notation for
    // decimal exponents -5 .. 15 and scientific otherwise, with an exponent of at least two digits
    const double ax = std::fabs(x);
    const bool want_sci = ax != 0.0 && (ax < 1e-4 || ax >= 1e16);
    const bool is_sci = out.find('e') != std::string::npos;
    if (want_sci != is_sci) {
        // re-derive the other notation from the digits and the exponent of the shortest scientific form
        std::string digits;
        int exp10 = 0;
        {
            char sb[40];
            const auto rs = std::to_chars(sb, sb + sizeof(sb), x, std::chars_format::scientific);
            std::string sci(sb, rs.ptr);  // d[.ddd]e[+-]dd, shortest
            const size_t e = sci.find('e');
            exp10 = std::stoi(sci.substr(e + 1));
            for (char ch : sci.substr(0, e))
                if (ch >= '0' && ch <= '9') digits.push_back(ch);
        }
        const bool neg = x < 0;
        std::string body;
        if (want_sci) {
            body = digits.substr(0, 1);
            if (digits.size() > 1) body += "." + digits.substr(1);
            char eb[16];
            std::snprintf(eb, sizeof(eb), "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
            body += eb;
        } else if (exp10 >= 0) {
            if ((int)digits.size() <= exp10 + 1) body = digits + std::string(exp10 + 1 - digits.size(), '0') + ".0";
            else body = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
        } else {
            body = "0." + std::string(-exp10 - 1, '0') + digits;
        }
        return (neg ? "-" : "") + body;
    }
    if (is_sci) {  // exponent with at least two digits
        const size_t e = out.find('e');
        std::string mant = out.substr(0, e), ex = out.substr(e + 1);
        const char sign = (ex[0] == '-' || ex[0] == '+') ? ex[0] : '+';
        if (ex[0] == '-' || ex[0] == '+') ex = ex.substr(1);
        if (ex.size() < 2) ex = "0" + ex;
        return mant + "e" + sign + ex;
    }
    if (out.find('.') == std::string::npos) out += ".0";
    return out;
}

inline std::string Quote(const std::string& s) {
    std::string out = "\"";
    for (unsigned char ch : s) {
        switch (ch) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (ch < 0x20) {
                    char b[8];
                    std::snprintf(b, sizeof(b), "\\u%04x", ch);
                    out += b;
                } else {
                    out.push_back((char)ch);
                }
        }
    }
    return out + "\"";
}

inline void Dump(const Json& j, std::string& out, int indent = 4, int level = 0) {  // std::setw(4) << json
    const std::string pad(indent * level, ' '), pad_in(indent * (level + 1), ' ');
    switch (j.kind) {
        case Json::Null: out += "null"; break;
        case Json::Bool: out += j.b ? "true" : "false"; break;
        case Json::Int: out += std::to_string(j.i); break;
        case Json::Float: out += FormatDouble(j.f); break;
        case Json::String: out += Quote(j.s); break;
        case Json::Array:
            if (j.a.empty()) { out += "[]"; break; }
            out += "[\n";
            for (size_t k = 0; k < j.a.size(); ++k) {
                out += pad_in;
                Dump(j.a[k], out, indent, level + 1);
                out += k + 1 < j.a.size() ? ",\n" : "\n";
            }
            out += pad + "]";
            break;
        case Json::Object: {
            if (j.o.empty()) { out += "{}"; break; }
            out += "{\n";
            size_t k = 0;
            for (const auto& [key, val] : j.o) {
                out += pad_in + Quote(key) + ": ";
                Dump(val, out, indent, level + 1);
                out += ++k < j.o.size() ? ",\n" : "\n";
            }
            out += pad + "}";
            break;
        }
    }
}
inline std::string Dump(const Json& j) {
    std::string out;
    Dump(j, out);
    return out;
}

// recursive-descent reader (objects, arrays, strings with the common escapes, numbers, literals)
class Reader {
public:
    explicit Reader(const std::string& text) : t_(text) {}
    Json Parse() {
        Json v = Value();
        Skip();
        if (p_ != t_.size()) Fail("trailing characters");
        return v;
    }

private:
    const std::string& t_;
    size_t p_ = 0;
    [[noreturn]] void Fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p_)); }
    void Skip() { while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) ++p_; }
    bool Lit(const char* w) {
        const size_t n = std::char_traits<char>::length(w);
        if (t_.compare(p_, n, w) == 0) { p_ += n; return true; }
        return false;
    }
    std::string Str() {
        if (t_[p_] != '"') Fail("expected a string");
        ++p_;
        std::string out;
        while (p_ < t_.size() && t_[p_] != '"') {
            char ch = t_[p_++];
            if (ch == '\\') {
                if (p_ >= t_.size()) Fail("bad escape");
                const char e = t_[p_++];
                switch (e) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (p_ + 4 > t_.size()) Fail("bad \\u escape");
                        const unsigned cp = (unsigned)std::stoul(t_.substr(p_, 4), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out.push_back((char)cp);
                        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    default: out.push_back(e);
                }
            } else {
                out.push_back(ch);
            }
        }
        if (p_ >= t_.size()) Fail("unterminated string");
        ++p_;
        return out;
    }
    Json Value() {
        Skip();
        if (p_ >= t_.size()) Fail("unexpected end");
        const char ch = t_[p_];
        if (ch == '{') {
            ++p_;
            Json j = Json::object();
            Skip();
            if (t_[p_] == '}') { ++p_; return j; }
            for (;;) {
                Skip();
                std::string key = Str();
                Skip();
                if (t_[p_] != ':') Fail("expected ':'");
                ++p_;
                j.o[key] = Value();
                Skip();
                if (t_[p_] == ',') { ++p_; continue; }
                if (t_[p_] == '}') { ++p_; return j; }
                Fail("expected ',' or '}'");
            }
        }
        if (ch == '[') {
            ++p_;
            Json j = Json::array();
            Skip();
            if (t_[p_] == ']') { ++p_; return j; }
            for (;;) {
                j.a.push_back(Value());
                Skip();
                if (t_[p_] == ',') { ++p_; continue; }
                if (t_[p_] == ']') { ++p_; return j; }
                Fail("expected ',' or ']'");
            }
        }
        if (ch == '"') return Json(Str());
        if (Lit("true")) return Json(true);
        if (Lit("false")) return Json(false);
        if (Lit("null")) return Json();
        const size_t start = p_;
        bool is_float = false;
        while (p_ < t_.size() && (std::isdigit((unsigned char)t_[p_]) || t_[p_] == '-' || t_[p_] == '+' || t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E')) {
            if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') is_float = true;
            ++p_;
        }
        if (p_ == start) Fail("unexpected character");
        const std::string num = t_.substr(start, p_ - start);
        if (is_float) return Json(std::stod(num));
        return Json((long long)std::stoll(num));
    }
};
inline Json ParseJson(const std::string& text) { return Reader(text).Parse(); }
inline Json LoadJson(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ParseJson(ss.str());
}

// ---- config ------------------------------------------------------------------------------------------
struct OptimizerOptions {  // OptimizerConfig, SfmConfig.h:15-22
    std::string method;
    int maxIter = 0;
    float maxTolerance = 0.f;
    float delta = 0.f;
    bool usePreconditioner = false;
};

struct SfmConfig {
    std::string imagesPath, outputTransformPath;
    int minFeaturesCount = 0, maxFeaturesCount = 0;
    float inliersRatio = 0.f;
    int maxDataSize = 0;
    unsigned initialMinInliers = 0;
    float initialMaxReprError = 0.f, initialMinTriAngle = 0.f, maxReprError = 0.f, minTriAngle = 0.f;
    int minPnpInliers = 0;
    OptimizerOptions refineOpt, globalOpt;
    bool ui = false;

    static SfmConfig Parse(const Json& data) {  // SfmConfig::Parse, field by field
        SfmConfig r;
        const std::string root = data.at("root_path").string();
        r.imagesPath = root + data.at("images_path").string();
        r.outputTransformPath = root + data.at("transform_path").string();
        r.maxDataSize = (int)data.at("max_data_count").integer();
        const Json& ui = data.at("ui");
        r.ui = ui.kind == Json::String && ui.s == "true";  // `data["ui"] == "true"` (:36): a JSON boolean compares unequal
        const Json& feature = data.at("feature");
        r.minFeaturesCount = (int)feature.at("min_features_count").integer();
        r.maxFeaturesCount = (int)feature.at("max_features_count").integer();
        r.inliersRatio = (float)feature.at("inliers_ratio").number();
        const Json& rec = data.at("reconstruction");
        const Json &ini = rec.at("initial_pair"), &proc = rec.at("processing");
        r.initialMinInliers = (unsigned)ini.at("min_inliers").integer();
        r.initialMaxReprError = (float)ini.at("max_reprojection_error").number();
        r.initialMinTriAngle = (float)ini.at("min_angle").number();
        r.initialMinTriAngle *= 3.141592 / 180.0;  // (:48-49) the literal and the float rounding of the reference
        r.maxReprError = (float)proc.at("max_reprojection_error").number();
        r.minTriAngle = (float)proc.at("min_angle").number();
        r.minTriAngle *= 3.141592 / 180.0;
        r.minPnpInliers = (int)proc.at("min_pnp_inliers").integer();
        const Json &refine = data.at("refine_ba"), &global = data.at("global_ba");
        r.refineOpt.maxIter = (int)refine.at("max_iter").integer();
        r.refineOpt.maxTolerance = (float)refine.at("max_toler").number();
        r.refineOpt.method = refine.at("method").string();
        r.refineOpt.delta = (float)refine.at("delta").number();
        r.refineOpt.usePreconditioner = refine.at("use_preconditioner").boolean();
        r.globalOpt.maxIter = (int)global.at("max_iter").integer();
        r.globalOpt.maxTolerance = (float)global.at("max_toler").number();
        r.globalOpt.method = global.at("method").string();
        r.globalOpt.delta = (float)refine.at("delta").number();                       // (:67) taken from refine_ba
        r.globalOpt.usePreconditioner = refine.at("use_preconditioner").boolean();    // (:68) idem
        return r;
    }
};

// ---- transform.json ------------------------------------------------------------------------------------
using Pose = std::array<double, 16>;  // row-major 4x4

inline Json MatrixJson(const Pose& T) {
    Json m = Json::array();
    for (int r = 0; r < 4; ++r) {
        Json row = Json::array();
        for (int c = 0; c < 4; ++c) row.push_back(Json(T[4 * r + c]));
        m.push_back(row);
    }
    return m;
}

inline Json PositionsDocument(const std::map<unsigned, std::pair<std::string, Pose>>& positions, const float w, const float h,
                              const float cx, const float cy, const float fx, const float fy) {
    Json frames;
    frames["version"] = Json(0);
    frames["w"] = Json((double)w);
    frames["h"] = Json((double)h);
    frames["cx"] = Json((double)cx);
    frames["cy"] = Json((double)cy);
    frames["fl_x"] = Json((double)fx);
    frames["fl_y"] = Json((double)fy);
    for (const char* k : {"k1", "k2", "k3", "k4", "p1", "p2"}) frames[k] = Json(0);
    frames["is_fisheye"] = Json(false);
    const float angleX = std::atan(w / (fx * 2.0)) * 2.0;
    const float angleY = std::atan(h / (fy * 2.0)) * 2.0;
    frames["camera_angle_x"] = Json((double)angleX);
    frames["camera_angle_y"] = Json((double)angleY);
    frames["fovx"] = Json(angleX * 180.0 / 3.141592);
    frames["fovy"] = Json(angleY * 180.0 / 3.141592);
    frames["frames"] = Json();  // `frames["frames"] = { }` is null until the first push_back
    for (const auto& [id, posTuple] : positions) {
        Json frame;
        frame["file_path"] = Json(posTuple.first);
        frame["transform_matrix"] = MatrixJson(posTuple.second);
        frames["frames"].push_back(frame);
    }
    return frames;
}

inline void SavePositions(const std::string& path, const std::map<unsigned, std::pair<std::string, Pose>>& positions, const float w,
                          const float h, const float cx, const float cy, const float fx, const float fy) {
    std::ofstream file(path, std::ios_base::out);
    if (file.is_open()) file << Dump(PositionsDocument(positions, w, h, cx, cy, fx, fy)) << std::endl;
}

inline Pose Inverse(const Pose& M) {  // general 4x4 inverse, Gauss-Jordan with partial pivoting
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 8; ++c) a[r][c] = c < 4 ? M[4 * r + c] : (c - 4 == r ? 1.0 : 0.0);
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) throw std::runtime_error("singular pose matrix");
        for (int c = 0; c < 8; ++c) std::swap(a[col][c], a[piv][c]);
        const double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
        }
    }
    Pose out;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = a[r][4 + c];
    return out;
}

inline Pose PoseToNerf(const Pose& T) {  // TransformToNerf.cpp:52-58: inverse(T) * diag(1, -1, -1, 1)
    Pose P = Inverse(T);
    for (int r = 0; r < 4; ++r) {
        P[4 * r + 1] = -P[4 * r + 1];
        P[4 * r + 2] = -P[4 * r + 2];
    }
    return P;
}

inline std::string TransformToNerf(std::string folder) {  // <folder>/transform.json -> <folder>/transforms_nerf.json
    if (folder.empty() || folder.back() != '/') folder += "/";
    Json doc = LoadJson(folder + "transform.json");
    Json& frames = doc["frames"];
    for (Json& frame : frames.a) {
        const Json& m = frame.at("transform_matrix");
        Pose T;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) T[4 * r + c] = m.a.at(r).a.at(c).number();
        frame["transform_matrix"] = MatrixJson(PoseToNerf(T));
    }
    const std::string out = folder + "transforms_nerf.json";
    std::ofstream file(out, std::ios_base::out);
    if (file.is_open()) file << Dump(doc) << std::endl;
    return out;
}

}  // namespace io
}  // namespace hip
}  // namespace eacham
