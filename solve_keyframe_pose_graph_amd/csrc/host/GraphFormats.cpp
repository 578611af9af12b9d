// GraphFormats.cpp — see GraphFormats.hpp.  Own implementation; the key names and value conventions follow the reference's writers
// (src/NodeDataManager.cpp:503-628, src/PoseGraphSLAM.cpp:1111-1207) so that files are interchangeable with its tools.
#include "GraphFormats.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pgo_host {

// ------------------------------------------------------------------------------------------------ JSON reader
namespace {
const JsonValue kNull;

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit Parser(const std::string& text) : s(text) {}

    bool fail(const char* what) { if (err.empty()) { err = std::string(what) + " at byte " + std::to_string(i); } return false; }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }

    bool string(std::string& out) {
        if (i >= s.size() || s[i] != '"') return fail("expected string");
        ++i;
        out.clear();
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c != '\\') { out.push_back(c); continue; }
            if (i >= s.size()) return fail("dangling escape");
            c = s[i++];
            switch (c) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (i + 4 > s.size()) return fail("short \\u escape");
                    unsigned cp = 0;
                    for (int k = 0; k < 4; ++k) {
                        const char h = s[i++];
                        cp <<= 4;
                        if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                        else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                        else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                        else return fail("bad \\u escape");
                    }
                    // UTF-8 encode the BMP code point (surrogate pairs are passed through as two 3-byte sequences)
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: return fail("unknown escape");
            }
        }
        if (i >= s.size()) return fail("unterminated string");
        ++i;
        return true;
    }

    bool value(JsonValue& v, int depth) {
        if (depth > 200) return fail("nesting too deep");
        ws();
        if (i >= s.size()) return fail("unexpected end");
        const char c = s[i];
        if (c == '{') {
            v.kind = JsonValue::Object;
            ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!string(key)) return false;
                ws();
                if (i >= s.size() || s[i] != ':') return fail("expected ':'");
                ++i;
                if (!value(v.obj[key], depth + 1)) return false;
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JsonValue::Array;
            ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JsonValue::String; return string(v.str); }
        if (s.compare(i, 4, "true") == 0) { v.kind = JsonValue::Bool; v.b = true; i += 4; return true; }
        if (s.compare(i, 5, "false") == 0) { v.kind = JsonValue::Bool; v.b = false; i += 5; return true; }
        if (s.compare(i, 4, "null") == 0) { v.kind = JsonValue::Null; i += 4; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* b = s.c_str() + i;
            char* e = nullptr;
            v.num = std::strtod(b, &e);
            if (e == b) return fail("bad number");
            v.kind = JsonValue::Number;
            i += (size_t)(e - b);
            return true;
        }
        return fail("unexpected character");
    }
};

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) { if (c == sep) { out.push_back(cur); cur.clear(); } else cur.push_back(c); }
    out.push_back(cur);
    return out;
}

std::string num(double v) { char b[40]; std::snprintf(b, sizeof(b), "%.17g", v); return b; }
}  // namespace

const JsonValue& JsonValue::at(const std::string& k) const {
    if (kind != Object) return kNull;
    const auto it = obj.find(k);
    return it == obj.end() ? kNull : it->second;
}
const JsonValue& JsonValue::at(size_t i) const { return (kind == Array && i < arr.size()) ? arr[i] : kNull; }

bool json_parse(const std::string& text, JsonValue& out, std::string* err) {
    Parser p(text);
    out = JsonValue();
    bool ok = p.value(out, 0);
    if (ok) { p.ws(); if (p.i != text.size()) ok = p.fail("trailing characters"); }
    if (!ok && err) *err = p.err;
    return ok;
}

std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            default:
                if (c < 0x20) { char b[8]; std::snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
                else o.push_back((char)c);
        }
    }
    return o;
}

// ------------------------------------------------------------------------------------------------ matrices as strings
std::string matrix4d_to_csv(const Matrix4d& M) {
    std::string out;
    for (int r = 0; r < 4; ++r) {
        for (int c = 0; c < 4; ++c) { out += num(M(r, c)); if (c < 3) out += ","; }
        if (r < 3) out += ";";
    }
    return out;
}

bool csv_to_matrix4d(const std::string& s, Matrix4d& M) {
    const std::vector<std::string> rows = split(s, ';');
    if (rows.size() != 4) return false;
    for (int r = 0; r < 4; ++r) {
        const std::vector<std::string> cols = split(rows[r], ',');
        if (cols.size() != 4) return false;
        for (int c = 0; c < 4; ++c) {
            char* e = nullptr;
            const double v = std::strtod(cols[c].c_str(), &e);
            if (e == cols[c].c_str()) return false;
            M(r, c) = v;
        }
    }
    return true;
}

std::string prettyprint_matrix4d(const Matrix4d& M) {
    const double nx = M(0, 0), ny = M(1, 0), nz = M(2, 0), ox = M(0, 1), oy = M(1, 1), ax = M(0, 2), ay = M(1, 2);
    const double y = std::atan2(ny, nx);
    const double p = std::atan2(-nz, nx * std::cos(y) + ny * std::sin(y));
    const double r = std::atan2(ax * std::sin(y) - ay * std::cos(y), -ox * std::sin(y) + oy * std::cos(y));
    char b[200];
    std::snprintf(b, sizeof(b), ":YPR(deg)=(%4.3f,%4.3f,%4.3f)  :TxTyTz=(%4.3f,%4.3f,%4.3f)", y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0, M(0, 3), M(1, 3), M(2, 3));
    return b;
}

// ------------------------------------------------------------------------------------------------ log_posegraph.json
std::string VectorGraphSource::disjoint_set_status() const {
    const int nw = n_worlds();
    if (nw > 0) ensure_world(nw - 1);
    std::map<int, std::string> members;
    std::string out;
    int set_count = 0;
    for (int w = 0; w < nw; ++w) if (set_of_[w] == w) ++set_count;
    out += "element_count=" + std::to_string(nw) + "   set_count=" + std::to_string(set_count) + ";";
    for (int w = 0; w < nw; ++w) {
        out += "world#" + std::to_string(w) + " is in setID=" + std::to_string(set_of_[w]) + ";";
        std::string& m = members[set_of_[w]];
        m += (m.empty() ? "" : ",") + std::to_string(w);
    }
    out += ";";
    for (const auto& kv : members) out += "set#" + std::to_string(kv.first) + " contains worlds: " + kv.second + ";";
    return out;
}

bool save_posegraph_json(const VectorGraphSource& src, const std::string& base_path) {
    std::ofstream f(base_path + "/log_posegraph.json");
    if (!f.is_open()) return false;
    const int N = src.getNodeLen(), E = src.getEdgeLen(), W = src.n_worlds();
    const std::string zero_cov = "0,0,0,0,0,0;0,0,0,0,0,0;0,0,0,0,0,0;0,0,0,0,0,0;0,0,0,0,0,0;0,0,0,0,0,0";   // the synthetic sources carry no covariance
    f << "{\n    \"meta_data\": {\"getNodeLen\": " << N << ", \"getEdgeLen\": " << E << ", \"n_worlds\": " << W << "},\n";
    f << "    \"nodes\": [";
    for (int i = 0; i < N; ++i) {
        const Matrix4d wTc = src.getNodePose(i);
        f << (i ? ",\n" : "\n") << "        {\"timestamp\": " << num(src.getNodeTimestamp(i)) << ", \"idx\": " << i << ", \"world_id\": " << src.which_world_is_this_node(i)
          << ", \"wTc\": \"" << matrix4d_to_csv(wTc) << "\", \"wTc_pretty\": \"" << json_escape(prettyprint_matrix4d(wTc)) << "\", \"cov\": \"" << zero_cov << "\"}";
    }
    f << "\n    ],\n    \"loopedges\": [";
    for (int e = 0; e < E; ++e) {
        const std::pair<int, int> p = src.getEdgeIdxInfo(e);
        const int w0 = src.which_world_is_this_node(p.first), w1 = src.which_world_is_this_node(p.second);
        const int code = (w0 < 0 || w1 < 0) ? -1 : (w0 == w1 ? 1 : 2);          // reference :560-566
        const Matrix4d bTa = src.getEdgePose(e);
        f << (e ? ",\n" : "\n") << "        {\"idx0\": " << p.first << ", \"idx1\": " << p.second << ", \"timestamp0\": " << num(src.getNodeTimestamp(p.first))
          << ", \"timestamp1\": " << num(src.getNodeTimestamp(p.second)) << ", \"world0_id\": " << w0 << ", \"world1_id\": " << w1 << ", \"code\": " << code
          << ", \"b_T_a\": \"" << matrix4d_to_csv(bTa) << "\", \"b_T_a_pretty\": \"" << json_escape(prettyprint_matrix4d(bTa)) << "\", \"weight\": " << num(src.getEdgeWeight(e))
          << ", \"description\": \"" << json_escape(src.getEdgeDescriptionString(e)) << "\"}";
    }
    f << "\n    ],\n    \"world_info\": [";
    for (int w = 0; w < W; ++w)
        f << (w ? ", " : "") << "{\"id\": " << w << ", \"nodeidx_of_world_i_started\": " << src.nodeidx_of_world_i_started(w) << ", \"nodeidx_of_world_i_ended\": " << src.nodeidx_of_world_i_ended(w) << "}";
    f << "],\n    \"kidnap_info\": [";
    for (int w = 0; w + 1 < W; ++w) {   // kidnap w separates world w from world w+1 (reference NodeDataManager.cpp:597-608)
        const int a = src.nodeidx_of_world_i_ended(w), b = src.nodeidx_of_world_i_started(w + 1);
        f << (w ? ", " : "") << "{\"idx\": " << w << ", \"stamp_of_kidnap_i_started\": " << num(a >= 0 ? src.getNodeTimestamp(a) : 0.0)
          << ", \"stamp_of_kidnap_i_ended\": " << num(b >= 0 ? src.getNodeTimestamp(b) : 0.0) << "}";
    }
    f << "],\n    \"disjoint_set_status\": \"" << json_escape(src.disjoint_set_status()) << "\"\n}\n";
    return f.good();
}

bool load_posegraph_json(VectorGraphSource& src, const std::string& base_path, const std::vector<bool>& edge_mask, std::string* err) {
    std::string dummy;
    std::string& e = err ? *err : dummy;
    std::ifstream f(base_path + "/log_posegraph.json");
    if (!f.is_open()) { e = "cannot open " + base_path + "/log_posegraph.json"; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    JsonValue all;
    if (!json_parse(ss.str(), all, &e)) return false;
    const JsonValue& nodes = all.at("nodes");
    const JsonValue& edges = all.at("loopedges");
    const int metaN = all.at("meta_data").at("getNodeLen").as_int(-1), metaE = all.at("meta_data").at("getEdgeLen").as_int(-1);
    if (metaN != (int)nodes.size() || metaE != (int)edges.size()) { e = "meta_data and the arrays are not consistent"; return false; }   // reference :659-666
    src.reset();
    for (size_t i = 0; i < nodes.size(); ++i) {
        const JsonValue& n = nodes.at(i);
        Matrix4d wTc;
        if (!csv_to_matrix4d(n.at("wTc").as_string(), wTc)) { e = "node " + std::to_string(i) + ": wTc is not a 4x4 matrix string"; src.reset(); return false; }
        // worlds are numbered in order of appearance (one more per kidnap): an id beyond the keyframe count cannot come from a recorded
        // session, and the world tables grow with the largest id
        const int world_id = n.has("world_id") ? n.at("world_id").as_int(0) : 0;
        if (world_id > (int)nodes.size() || world_id < -(int)nodes.size() - 1) { e = "node " + std::to_string(i) + ": world_id out of range"; src.reset(); return false; }
        src.add_node(world_id, wTc, n.at("timestamp").as_double((double)i * 0.1));
    }
    for (size_t k = 0; k < edges.size(); ++k) {
        if (!edge_mask.empty() && (k >= edge_mask.size() || !edge_mask[k])) continue;                                       // reference :700-701
        const JsonValue& ed = edges.at(k);
        const int idx0 = ed.at("idx0").as_int(-1), idx1 = ed.at("idx1").as_int(-1);
        if (idx0 < 0 || idx1 < 0 || idx0 >= (int)nodes.size() || idx1 >= (int)nodes.size()) { e = "loop edge " + std::to_string(k) + ": endpoint out of range"; src.reset(); return false; }
        if (idx0 == idx1) { e = "loop edge " + std::to_string(k) + ": both endpoints are keyframe " + std::to_string(idx0); src.reset(); return false; }
        // the claimed stamps must be the stamps of the keyframes (reference :736-747 exit(1)s otherwise)
        if (ed.has("timestamp0") && ed.at("timestamp0").as_double() != src.getNodeTimestamp(idx0)) { e = "loop edge " + std::to_string(k) + ": timestamp0 differs from its keyframe's"; src.reset(); return false; }
        if (ed.has("timestamp1") && ed.at("timestamp1").as_double() != src.getNodeTimestamp(idx1)) { e = "loop edge " + std::to_string(k) + ": timestamp1 differs from its keyframe's"; src.reset(); return false; }
        Matrix4d bTa;
        if (!csv_to_matrix4d(ed.at("b_T_a").as_string(), bTa)) { e = "loop edge " + std::to_string(k) + ": b_T_a is not a 4x4 matrix string"; src.reset(); return false; }
        src.add_loop_edge(idx0, idx1, bTa, ed.at("weight").as_double(1.0), ed.at("description").as_string());
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ solved_posegraph.json
namespace {
std::string matrix_json(const Matrix4d& M) {   // RawFileIO::eigen_matrix_to_json: rows/cols/data with IOFormat(FullPrecision, DontAlignCols, ", ", "\n")
    std::string data;
    for (int r = 0; r < 4; ++r) {
        for (int c = 0; c < 4; ++c) { data += num(M(r, c)); if (c < 3) data += ", "; }
        if (r < 3) data += "\n";
    }
    return "{\"rows\": 4, \"cols\": 4, \"data\": \"" + json_escape(data) + "\", \"data_pretty\": \"" + json_escape(prettyprint_matrix4d(M)) + "\"}";
}
bool matrix_from_json(const JsonValue& v, Matrix4d& M) {   // RawFileIO::read_eigen_matrix4d_fromjson (src/utils/RawFileIO.cpp:372-409)
    if (v.at("rows").as_int(0) != 4 || v.at("cols").as_int(0) != 4) return false;
    const std::vector<std::string> rows = split(v.at("data").as_string(), '\n');
    if (rows.size() != 4) return false;
    for (int r = 0; r < 4; ++r) {
        const std::vector<std::string> cols = split(rows[r], ',');
        if (cols.size() != 4) return false;
        for (int c = 0; c < 4; ++c) { char* e = nullptr; M(r, c) = std::strtod(cols[c].c_str(), &e); if (e == cols[c].c_str()) return false; }
    }
    return true;
}
long long to_nsec(double stamp_s) { return (long long)std::llround(stamp_s * 1e9); }
}  // namespace

bool save_solved_posegraph_json(const VectorGraphSource& src, const std::vector<Matrix4d>& w_T_c, const std::string& base_path) {
    const int N = src.getNodeLen(), W = src.n_worlds();
    if ((int)w_T_c.size() != N) return false;
    std::ofstream f(base_path + "/solved_posegraph.json");
    if (!f.is_open()) return false;
    f << "{\n    \"SolvedPoseGraph\": [";
    for (int i = 0; i < N; ++i) {
        const int w = src.which_world_is_this_node(i);
        f << (i ? ",\n" : "\n") << "        {\"w_T_c\": " << matrix_json(w_T_c[i]) << ", \"worldID\": " << w << ", \"setID_of_worldID\": " << src.find_setID_of_world_i(w)
          << ", \"stampNSec\": " << to_nsec(src.getNodeTimestamp(i)) << ", \"seq\": " << i << "}";
    }
    f << "\n    ],\n    \"KidnapTimestamps\": {\"kidnap_starts\": [";
    for (int w = 0; w + 1 < W; ++w) { const int a = src.nodeidx_of_world_i_ended(w); f << (w ? ", " : "") << "{\"stampNSec\": " << to_nsec(a >= 0 ? src.getNodeTimestamp(a) : 0.0) << "}"; }
    f << "], \"kidnap_ends\": [";
    for (int w = 0; w + 1 < W; ++w) { const int b = src.nodeidx_of_world_i_started(w + 1); f << (w ? ", " : "") << "{\"stampNSec\": " << to_nsec(b >= 0 ? src.getNodeTimestamp(b) : 0.0) << "}"; }
    f << "]},\n    \"WorldsData\": {\n        \"rel_pose_between_worlds__wb_T_wa\": [";
    std::string log, debug;
    bool first = true;
    for (int w = 0; w < W; ++w) { log += "add_element:" + std::to_string(w) + ";"; debug += "\t\t\tadd_element( " + std::to_string(w) + ")\n"; }
    for (int w = 0; w < W; ++w) {
        const int root = src.find_setID_of_world_i(w);
        if (root < 0 || root == w) continue;
        f << (first ? "\n" : ",\n") << "            {\"node_b\": " << root << ", \"node_a\": " << w << ", \"wb_T_wa\": " << matrix_json(src.getPoseBetweenWorlds(root, w))
          << ", \"info_wb_T_wa\": \"merged\"}";
        first = false;
        log += "union_sets:" + std::to_string(std::max(w, root)) + "," + std::to_string(std::min(w, root)) + ";";
        debug += "\t\t\tunion_sets( " + std::to_string(std::max(w, root)) + "," + std::to_string(std::min(w, root)) + ")\n";
    }
    f << (first ? "" : "\n        ") << "],\n        \"vec_world_starts\": [";
    for (int w = 0; w < W; ++w) { const int a = src.nodeidx_of_world_i_started(w); f << (w ? ", " : "") << "{\"stampNSec\": " << to_nsec(a >= 0 ? src.getNodeTimestamp(a) : 0.0) << "}"; }
    f << "],\n        \"vec_world_ends\": [";
    for (int w = 0; w < W; ++w) { const int a = src.nodeidx_of_world_i_ended(w); f << (w ? ", " : "") << "{\"stampNSec\": " << to_nsec(a >= 0 ? src.getNodeTimestamp(a) : 0.0) << "}"; }
    f << "],\n        \"disjoint_set\": {\"debug_string\": \"" << json_escape(debug) << "\", \"log_string\": \"" << json_escape(log) << "\"}\n    }\n}\n";
    return f.good();
}

bool load_solved_posegraph_json(VectorGraphSource& src, std::vector<Matrix4d>& w_T_c, const std::string& base_path, std::string* err) {
    std::string dummy;
    std::string& e = err ? *err : dummy;
    std::ifstream f(base_path + "/solved_posegraph.json");
    if (!f.is_open()) { e = "cannot open " + base_path + "/solved_posegraph.json"; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    JsonValue all;
    if (!json_parse(ss.str(), all, &e)) return false;
    const JsonValue& nodes = all.at("SolvedPoseGraph");
    if (nodes.kind != JsonValue::Array) { e = "SolvedPoseGraph missing"; return false; }
    src.reset();
    w_T_c.clear();
    for (size_t i = 0; i < nodes.size(); ++i) {
        const JsonValue& n = nodes.at(i);
        Matrix4d T;
        if (!matrix_from_json(n.at("w_T_c"), T)) { e = "SolvedPoseGraph[" + std::to_string(i) + "].w_T_c is not a 4x4 matrix"; src.reset(); w_T_c.clear(); return false; }
        if (n.at("seq").as_int((int)i) != (int)i) { e = "SolvedPoseGraph is not in seq order"; src.reset(); w_T_c.clear(); return false; }
        const int world_id = n.at("worldID").as_int(0);
        if (world_id > (int)nodes.size() || world_id < -(int)nodes.size() - 1) { e = "SolvedPoseGraph[" + std::to_string(i) + "]: worldID out of range"; src.reset(); w_T_c.clear(); return false; }
        src.add_node(world_id, T, n.at("stampNSec").as_double(0.0) * 1e-9);
        w_T_c.push_back(T);
    }
    // world merges: replay the union log with the stored relative poses (Worlds::loadStateFromDisk, reference src/Worlds.cpp:519-667)
    const JsonValue& wd = all.at("WorldsData");
    const JsonValue& rel = wd.at("rel_pose_between_worlds__wb_T_wa");
    std::map<std::pair<int, int>, Matrix4d> poses;
    for (size_t k = 0; k < rel.size(); ++k) {
        Matrix4d T;
        if (!matrix_from_json(rel.at(k).at("wb_T_wa"), T)) { e = "rel_pose_between_worlds__wb_T_wa[" + std::to_string(k) + "] is not a 4x4 matrix"; src.reset(); w_T_c.clear(); return false; }
        poses[{rel.at(k).at("node_b").as_int(-1), rel.at(k).at("node_a").as_int(-1)}] = T;
    }
    for (const std::string& cmd : split(wd.at("disjoint_set").at("log_string").as_string(), ';')) {
        if (cmd.size() < 4) continue;
        const std::vector<std::string> sp = split(cmd, ':');
        if (sp.size() != 2 || (sp[0] != "add_element" && sp[0] != "union_sets")) { e = "unknown disjoint-set command `" + cmd + "`"; src.reset(); w_T_c.clear(); return false; }
        if (sp[0] == "add_element") continue;                  // worlds exist as soon as a keyframe names them
        const std::vector<std::string> ops = split(sp[1], ',');
        if (ops.size() != 2) { e = "union_sets needs two operands: `" + cmd + "`"; src.reset(); w_T_c.clear(); return false; }
        const int x = std::atoi(ops[0].c_str()), y = std::atoi(ops[1].c_str());
        const int b = std::min(x, y), a = std::max(x, y);
        auto it = poses.find({b, a});
        if (it != poses.end()) src.setPoseBetweenWorlds(b, a, it->second);
        else if ((it = poses.find({a, b})) != poses.end()) src.setPoseBetweenWorlds(a, b, it->second);
        else { e = "no relative pose stored for the merged worlds " + std::to_string(b) + " and " + std::to_string(a); src.reset(); w_T_c.clear(); return false; }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ g2o
bool export_g2o(const std::string& path, const std::vector<Matrix4d>& poses, const std::vector<G2oEdge>& edges) {
    std::FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return false;
    for (size_t i = 0; i < poses.size(); ++i) {
        double q[4], t[3];
        eigenmat_to_raw_xyzw(poses[i], q, t);
        std::fprintf(f, "VERTEX_SE3:QUAT %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
    }
    for (const G2oEdge& e : edges) {
        double q[4], t[3];
        eigenmat_to_raw_xyzw(e.c1_T_c2, q, t);
        std::fprintf(f, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", e.c1, e.c2, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
        const double wt = e.weight * e.weight, wr = 4.0 * e.weight * e.weight;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) std::fprintf(f, " %.17g", r == c ? (r < 3 ? wt : wr) : 0.0);
        std::fprintf(f, "\n");
    }
    const bool ok = std::ferror(f) == 0;
    std::fclose(f);
    return ok;
}

}  // namespace pgo_host
