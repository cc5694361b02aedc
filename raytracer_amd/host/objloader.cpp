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
