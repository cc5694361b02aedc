// objloader.cpp — see objloader.h.  A hand-rolled cursor tokenizer instead of the
// reference's nested istringstream/getline calls, with the same observable splitting
// rules: a token ends at the next single delimiter character; an exhausted line yields
// "no token" (which the numeric conversions read as 0, as atof("") does).
#include "objloader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

struct Cursor {
    const char *p, *end;
    // Extracts up to (and consuming) `delim`.  False only when nothing is left.
    bool next(char delim, std::string &tok)
    {
        tok.clear();
        if (p >= end) return false;
        const char *q = p;
        while (q < end && *q != delim) q++;
        tok.assign(p, q);
        p = q < end ? q + 1 : end;
        return true;
    }
};

float number(Cursor &c)
{
    std::string t;
    c.next(' ', t);
    return (float)atof(t.c_str());
}

}  // namespace

int LoadModel(const char *objFile, std::vector<vec3> &vertices, std::vector<vec2> &texcoords,
              std::vector<vec3> &normals, std::vector<int> &indices)
{
    FILE *f = fopen(objFile, "rb");
    if (!f) return 0;
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);

    int malformed = 0;
    const char *p = text.data(), *fend = p + text.size();
    std::string key, corner, pos;
    while (p < fend) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(fend - p));
        if (!eol) eol = fend;
        Cursor line{ p, eol };
        p = eol < fend ? eol + 1 : fend;
        line.next(' ', key);
        if (key == "v") {
            float x = number(line), y = number(line), z = number(line);
            vertices.push_back(vec3(x, y, z));
        } else if (key == "vt") {
            float u = number(line), v = number(line);
            texcoords.push_back(vec2(u, v));
        } else if (key == "vn") {
            float x = number(line), y = number(line), z = number(line);
            normals.push_back(vec3(x, y, z));
        } else if (key == "f") {
            int q[4];
            for (int i = 0; i < 4; i++) {
                line.next(' ', corner);
                Cursor c{ corner.data(), corner.data() + corner.size() };
                if (c.next('/', pos)) {
                    q[i] = atoi(pos.c_str()) - 1;   // OBJ counts from 1
                } else {
                    if (i != 3) malformed++;        // only the 4th corner may be absent
                    q[i] = -1;
                }
            }
            indices.push_back(q[0]); indices.push_back(q[1]); indices.push_back(q[2]);
            if (q[3] != -1) { indices.push_back(q[0]); indices.push_back(q[2]); indices.push_back(q[3]); }
        }
    }
    return malformed;
}

// ------------------------------------------------------------------ tolerant parser with materials
namespace {

// Next whitespace-delimited token of [p, end); false at the end of the line.
bool token(const char *&p, const char *end, std::string &tok)
{
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
    if (p >= end || *p == '#') return false;
    const char *q = p;
    while (q < end && *q != ' ' && *q != '\t' && *q != '\r') q++;
    tok.assign(p, q);
    p = q;
    return true;
}

bool read_floats(const char *&p, const char *end, float *dst, int n)
{
    std::string t;
    for (int i = 0; i < n; i++) {
        if (!token(p, end, t)) return false;
        dst[i] = (float)atof(t.c_str());
    }
    return true;
}

ObjMaterial default_material(const char *name)
{
    ObjMaterial m;
    memset(&m, 0, sizeof m);
    snprintf(m.name, sizeof m.name, "%s", name);
    m.reflect = 0.0f; m.diffuse = 1.0f; m.spec = 1.0f;         // Material(), main.cpp:34
    m.color[0] = m.color[1] = m.color[2] = 1.0f;
    return m;
}

bool read_file(const std::string &path, std::string &text)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    text.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    return true;
}

void load_mtl(const std::string &path, std::vector<ObjMaterial> &mats)
{
    std::string text;
    if (!read_file(path, text)) return;
    const char *p = text.data(), *fend = p + text.size();
    int cur = -1;
    bool has_ks = false, has_reflect = false;
    int illum = -1;
    float ks_max = 1.0f;
    auto finish = [&]() {
        if (cur < 0) return;
        if (!has_reflect && illum >= 3 && illum <= 7) mats[(size_t)cur].reflect = has_ks ? ks_max : 1.0f;
    };
    std::string key, t;
    while (p < fend) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(fend - p));
        if (!eol) eol = fend;
        const char *q = p;
        p = eol < fend ? eol + 1 : fend;
        if (!token(q, eol, key)) continue;
        if (key == "newmtl") {
            finish();
            if (!token(q, eol, t)) t = "";
            mats.push_back(default_material(t.c_str()));
            cur = (int)mats.size() - 1;
            has_ks = has_reflect = false; illum = -1; ks_max = 1.0f;
        } else if (cur >= 0 && key == "Kd") {
            float c[3];
            if (read_floats(q, eol, c, 3)) { mats[(size_t)cur].color[0] = c[0]; mats[(size_t)cur].color[1] = c[1]; mats[(size_t)cur].color[2] = c[2]; }
        } else if (cur >= 0 && key == "Ks") {
            float c[3];
            if (read_floats(q, eol, c, 3)) {
                ks_max = c[0] > c[1] ? (c[0] > c[2] ? c[0] : c[2]) : (c[1] > c[2] ? c[1] : c[2]);
                mats[(size_t)cur].spec = ks_max;
                has_ks = true;
            }
        } else if (cur >= 0 && key == "illum") {
            float v;
            if (read_floats(q, eol, &v, 1)) illum = (int)v;
        } else if (cur >= 0 && key == "reflect") {
            float v;
            if (read_floats(q, eol, &v, 1)) { mats[(size_t)cur].reflect = v; has_reflect = true; }
        }
    }
    finish();
}

}  // namespace

bool LoadModelEx(const char *objFile, ObjModel &out)
{
    std::string text;
    if (!read_file(objFile, text)) return false;
    std::string dir(objFile);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);

    out.materials.clear();
    out.materials.push_back(default_material(""));
    int cur_mat = 0;
    const char *p = text.data(), *fend = p + text.size();
    std::string key, t;
    std::vector<int> corners, corner_vt, corner_vn;
    // a `v/vt/vn` reference: 1-based, negative = relative to the elements read so far, absent or 0 = none (-1)
    auto side_index = [](const char *&cur, size_t count) -> int {
        if (*cur != '/') return -1;
        cur++;
        char *endp = nullptr;
        long v = strtol(cur, &endp, 10);
        if (endp == cur) return -1;
        cur = endp;
        long idx = v > 0 ? v - 1 : (v < 0 ? (long)count + v : -1);
        return idx < 0 || idx > 0x7ffffffe ? -1 : (int)idx;
    };
    while (p < fend) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(fend - p));
        if (!eol) eol = fend;
        const char *q = p;
        p = eol < fend ? eol + 1 : fend;
        if (!token(q, eol, key)) continue;
        if (key == "v") {
            float c[3] = { 0.f, 0.f, 0.f };
            read_floats(q, eol, c, 3);
            out.vertices.push_back(vec3(c[0], c[1], c[2]));
        } else if (key == "vt") {
            float c[2] = { 0.f, 0.f };
            read_floats(q, eol, c, 2);
            out.texcoords.push_back(vec2(c[0], c[1]));
        } else if (key == "vn") {
            float c[3] = { 0.f, 0.f, 0.f };
            read_floats(q, eol, c, 3);
            out.normals.push_back(vec3(c[0], c[1], c[2]));
        } else if (key == "mtllib") {
            while (token(q, eol, t)) load_mtl(dir + t, out.materials);
        } else if (key == "usemtl") {
            cur_mat = 0;
            if (token(q, eol, t)) {
                bool found = false;
                for (size_t i = 1; i < out.materials.size(); i++)
                    if (t == out.materials[i].name) { cur_mat = (int)i; found = true; break; }
                if (!found) out.missing_materials++;
            }
        } else if (key == "f") {
            corners.clear(); corner_vt.clear(); corner_vn.clear();
            bool ok = true;
            while (token(q, eol, t)) {
                char *endp = nullptr;
                long v = strtol(t.c_str(), &endp, 10);          // position index = text before the first '/'
                if (endp == t.c_str() || v == 0) { ok = false; break; }
                long idx = v > 0 ? v - 1 : (long)out.vertices.size() + v;   // negative: relative to the vertices so far
                if (idx < 0 || idx > 0x7ffffffe) { ok = false; break; }      // positive ones may point ahead: checked at the end
                corners.push_back((int)idx);
                const char *side = endp;
                corner_vt.push_back(side_index(side, out.texcoords.size()));
                corner_vn.push_back(side_index(side, out.normals.size()));
            }
            if (!ok || corners.size() < 3) { out.dropped_faces++; continue; }
            for (size_t k = 1; k + 1 < corners.size(); k++) {   // fan around corner 0, as objloader.cpp:90-99 does for quads
                out.indices.push_back(corners[0]); out.indices.push_back(corners[k]); out.indices.push_back(corners[k + 1]);
                out.normal_indices.push_back(corner_vn[0]); out.normal_indices.push_back(corner_vn[k]); out.normal_indices.push_back(corner_vn[k + 1]);
                out.texcoord_indices.push_back(corner_vt[0]); out.texcoord_indices.push_back(corner_vt[k]); out.texcoord_indices.push_back(corner_vt[k + 1]);
                out.tri_material.push_back(cur_mat);
            }
        }
    }
    // a positive index may name a vertex that comes later in the file; now every vertex is known
    size_t kept = 0;
    const int nv = (int)out.vertices.size();
    for (size_t tri = 0; tri < out.tri_material.size(); tri++) {
        const int a = out.indices[3 * tri], b = out.indices[3 * tri + 1], c = out.indices[3 * tri + 2];
        if (a >= nv || b >= nv || c >= nv) { out.dropped_faces++; continue; }
        out.indices[3 * kept] = a; out.indices[3 * kept + 1] = b; out.indices[3 * kept + 2] = c;
        for (int k = 0; k < 3; k++) {
            const int vn = out.normal_indices[3 * tri + k], vt = out.texcoord_indices[3 * tri + k];
            out.normal_indices[3 * kept + k] = vn < (int)out.normals.size() ? vn : -1;        // may have pointed ahead, too
            out.texcoord_indices[3 * kept + k] = vt < (int)out.texcoords.size() ? vt : -1;
        }
        out.tri_material[kept++] = out.tri_material[tri];
    }
    out.indices.resize(3 * kept);
    out.normal_indices.resize(3 * kept);
    out.texcoord_indices.resize(3 * kept);
    out.tri_material.resize(kept);
    return true;
}

int CornerNormals(const ObjModel &m, std::vector<float> &out)
{
    const size_t n_tris = m.indices.size() / 3;
    out.assign(9 * n_tris, 0.0f);
    int with = 0;
    for (size_t t = 0; t < n_tris; t++) {
        const int a = m.normal_indices[3 * t], b = m.normal_indices[3 * t + 1], c = m.normal_indices[3 * t + 2];
        if (a < 0 || b < 0 || c < 0) continue;
        const int ix[3] = { a, b, c };
        for (int k = 0; k < 3; k++) {
            const vec3 &n = m.normals[(size_t)ix[k]];
            out[9 * t + 3 * k] = n.x; out[9 * t + 3 * k + 1] = n.y; out[9 * t + 3 * k + 2] = n.z;
        }
        with++;
    }
    return with;
}

int CornerTexcoords(const ObjModel &m, std::vector<float> &out)
{
    const size_t n_tris = m.indices.size() / 3;
    out.assign(6 * n_tris, 0.0f);
    int with = 0;
    for (size_t t = 0; t < n_tris; t++) {
        const int ix[3] = { m.texcoord_indices[3 * t], m.texcoord_indices[3 * t + 1], m.texcoord_indices[3 * t + 2] };
        if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0) continue;
        for (int k = 0; k < 3; k++) {
            const vec2 &c = m.texcoords[(size_t)ix[k]];
            out[6 * t + 2 * k] = c.x; out[6 * t + 2 * k + 1] = c.y;
        }
        with++;
    }
    return with;
}
