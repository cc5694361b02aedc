// tests/objfuzz.cpp DIR N: the OBJ loader behind its C ABI (include/rt_objhost.h) on N made-up files - vertex lines, faces of
// three to five corners with indices in and out of range and every slash form, and lines of OBJ / MTL fragments in any order,
// some files cut off anywhere - with both tokenizers.  Built by tests/test_host_cpu.py with the address and
// undefined-behaviour sanitizers; it checks that the loader neither crashes nor hands out a material index or (relaxed
// tokenizer) a vertex index outside what it loaded.
#include "rt_objhost.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
static unsigned long long s = 88172645463325252ull;
static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
static const char *frag[] = { "v ", "vt ", "vn ", "f ", "usemtl ", "mtllib ", "o ", "g ", "s ", "#", "newmtl ", "Kd ", "Ks ", "illum ", "\n", "\r\n", " ", "\t", "/", "//", "-", ".", "e", "1", "2", "3", "99999999999", "-1", "0", "1.5", "nan", "inf", "1e40", "a.mtl", "x", "" };
int main(int argc, char **argv) {
    if (argc < 3) return 64;
    const std::string dir = argv[1];
    const std::string obj = dir + "/in.obj", mtl = dir + "/a.mtl";
    int n = atoi(argv[2]);
    long long tris = 0, bad = 0;
    for (int it = 0; it < n; it++) {
        std::string text;
        int lines = rnd() % 60;
        int nv = 0;
        for (int l = 0; l < lines; l++) {
            unsigned kind = rnd() % 10;
            char buf[256];
            if (kind < 3) { snprintf(buf, sizeof buf, "v %g %g %g\n", (rnd() % 2000) / 100.0 - 10, (rnd() % 2000) / 100.0 - 10, (rnd() % 2000) / 100.0 - 10); text += buf; nv++; }
            else if (kind < 5 && nv > 0) {
                int k = 3 + rnd() % 3; text += "f";
                for (int c = 0; c < k; c++) { int idx = (int)(rnd() % (nv + 3)) - 1; if (rnd() % 4 == 0) snprintf(buf, sizeof buf, " %d/%d/%d", idx, (int)(rnd() % 5), (int)(rnd() % 5)); else if (rnd() % 5 == 0) snprintf(buf, sizeof buf, " %d//%d", idx, (int)(rnd() % 5)); else snprintf(buf, sizeof buf, " %d", idx); text += buf; }
                text += "\n";
            } else { int k = 1 + rnd() % 8; for (int c = 0; c < k; c++) text += frag[rnd() % (sizeof frag / sizeof *frag)]; text += "\n"; }
        }
        if (rnd() % 7 == 0) text.resize(text.size() * (rnd() % 100) / 100);      // cut anywhere, no final newline
        FILE *f = fopen(obj.c_str(), "wb"); fwrite(text.data(), 1, text.size(), f); fclose(f);
        if (rnd() % 3 == 0) { f = fopen(mtl.c_str(), "wb"); std::string m; int k = rnd() % 12; for (int c = 0; c < k; c++) { m += frag[rnd() % (sizeof frag / sizeof *frag)]; m += frag[rnd() % (sizeof frag / sizeof *frag)]; m += "\n"; } fwrite(m.data(), 1, m.size(), f); fclose(f); }
        for (int relaxed = 0; relaxed < 2; relaxed++) {
            rt_obj *o = rt_obj_load(obj.c_str(), relaxed);
            if (!o) { printf("NULL\n"); return 1; }
            int32_t c[4]; rt_obj_counts(o, c);
            std::vector<float> v(3 * (size_t)c[0] + 1), m(6 * (size_t)c[2] + 1);
            std::vector<int32_t> idx(c[1] + 1), tm(c[1] / 3 + 1);
            rt_obj_copy(o, v.data(), idx.data(), tm.data(), m.data());
            for (int k = 0; k < c[1]; k++) if (idx[k] < 0 || idx[k] >= c[0]) { if (relaxed) { printf("relaxed loader let index %d of %d vertices through (iteration %d)\n", idx[k], c[0], it); return 2; } }
            for (int k = 0; k < c[1] / 3; k++) if (tm[k] < 0 || tm[k] >= c[2]) { printf("material index %d of %d (iteration %d, relaxed %d)\n", tm[k], c[2], it, relaxed); return 3; }
            tris += c[1] / 3; bad += c[3];
            rt_obj_free(o);
        }
    }
    printf("%d files, %lld triangles, %lld faces dropped: no crash, no index out of range\n", n, tris, bad);
    return 0;
}
