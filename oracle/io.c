/*
 * oracle/io.c -- CPU restatement of the on-disk formats of the hot path.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).
 *
 * File names: /root/reference/src/R3DProject.cpp:854-871 (sfm_data.bin, matches.putative.txt,
 * matches.f.txt, matches.e.txt, matches.h.txt).  .feat / .desc: /root/reference/src/keypointSet.hpp:49-67
 * (saveFeatsToFile / saveDescsToBinFile are OpenMVG's: one "x y scale orientation" text line per
 * feature; a std::size_t count followed by the raw row-major descriptor bytes).
 * matches.*: OpenMVG matching/indMatch_utils.cpp Save/Load(PairWiseMatches) -- external, restated
 * from SURVEY.md A.7: ".txt" = "I J\n count\n" + count lines "i j\n"; ".bin" = cereal
 * PortableBinaryOutputArchive of std::map<std::pair<u32,u32>, std::vector<IndMatch>>:
 * 1 endianness byte (1 on little-endian hosts), u64 map size, then per entry
 * u32 I, u32 J, u64 count, count x (u32 i, u32 j).
 */
#include "r3d_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int has_ext(const char* path, const char* ext)
{
    const size_t lp = strlen(path), le = strlen(ext);
    return lp >= le && strcmp(path + lp - le, ext) == 0;
}

int orc_save_matches(const char* path, int64_t n_pairs, const uint32_t* pairs,
                     const uint32_t* counts, const orc_match* matches)
{
    const int bin = has_ext(path, ".bin");
    if (!bin && !has_ext(path, ".txt")) return -2;
    FILE* f = fopen(path, bin ? "wb" : "w");
    if (!f) return -1;
    int64_t off = 0;
    if (bin) {
        const uint8_t le = 1;
        uint64_t np = 0;
        for (int64_t p = 0; p < n_pairs; ++p) if (counts[p]) ++np;   /* empty entries never enter the map */
        fwrite(&le, 1, 1, f);
        fwrite(&np, 8, 1, f);
        for (int64_t p = 0; p < n_pairs; ++p) {
            if (counts[p]) {
                const uint64_t c = counts[p];
                fwrite(&pairs[2 * p], 4, 1, f);
                fwrite(&pairs[2 * p + 1], 4, 1, f);
                fwrite(&c, 8, 1, f);
                fwrite(matches + off, sizeof(orc_match), counts[p], f);
            }
            off += counts[p];
        }
    } else {
        for (int64_t p = 0; p < n_pairs; ++p) {
            if (counts[p]) {
                fprintf(f, "%u %u\n%u\n", pairs[2 * p], pairs[2 * p + 1], counts[p]);
                for (uint32_t k = 0; k < counts[p]; ++k)
                    fprintf(f, "%u %u\n", matches[off + k].i, matches[off + k].j);
            }
            off += counts[p];
        }
    }
    return fclose(f) == 0 ? 0 : -1;
}

int orc_load_matches(const char* path, int64_t* n_pairs, int64_t* n_matches,
                     uint32_t* pairs, uint32_t* counts, orc_match* matches)
{
    const int bin = has_ext(path, ".bin");
    if (!bin && !has_ext(path, ".txt")) return -2;
    FILE* f = fopen(path, bin ? "rb" : "r");
    if (!f) return -1;
    int64_t np = 0, nm = 0;
    if (bin) {
        uint8_t le; uint64_t cnt;
        if (fread(&le, 1, 1, f) != 1 || le != 1 || fread(&cnt, 8, 1, f) != 1) { fclose(f); return -3; }
        for (uint64_t p = 0; p < cnt; ++p) {
            uint32_t ij[2]; uint64_t c;
            if (fread(ij, 4, 2, f) != 2 || fread(&c, 8, 1, f) != 1) { fclose(f); return -3; }
            if (pairs) { pairs[2 * np] = ij[0]; pairs[2 * np + 1] = ij[1]; counts[np] = (uint32_t)c; }
            if (matches) { if (fread(matches + nm, sizeof(orc_match), c, f) != c) { fclose(f); return -3; } }
            else fseek(f, (long)(c * sizeof(orc_match)), SEEK_CUR);
            ++np; nm += (int64_t)c;
        }
    } else {
        unsigned I, J, c;
        while (fscanf(f, "%u %u %u", &I, &J, &c) == 3) {
            if (pairs) { pairs[2 * np] = I; pairs[2 * np + 1] = J; counts[np] = c; }
            for (unsigned k = 0; k < c; ++k) {
                unsigned a, b;
                if (fscanf(f, "%u %u", &a, &b) != 2) { fclose(f); return -3; }
                if (matches) { matches[nm + k].i = a; matches[nm + k].j = b; }
            }
            ++np; nm += c;
        }
    }
    fclose(f);
    *n_pairs = np; *n_matches = nm;
    return 0;
}

int orc_save_feat(const char* path, int n, const float* xyso)
{
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    /* operator<<(SIOPointFeature): "x y scale orientation", default ostream float formatting (%g, 6 digits) */
    for (int k = 0; k < n; ++k)
        fprintf(f, "%g %g %g %g\n", xyso[4 * k], xyso[4 * k + 1], xyso[4 * k + 2], xyso[4 * k + 3]);
    return fclose(f) == 0 ? 0 : -1;
}

int orc_load_feat(const char* path, int* n, float* xyso, int cap)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int k = 0; float v[4];
    while (fscanf(f, "%f %f %f %f", &v[0], &v[1], &v[2], &v[3]) == 4) {
        if (xyso && k < cap) memcpy(xyso + 4 * k, v, sizeof(v));
        ++k;
    }
    fclose(f);
    *n = k;
    return 0;
}

int orc_save_desc(const char* path, uint64_t n, size_t row_bytes, const void* data)
{
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(&n, 8, 1, f);                      /* std::size_t cardDesc (8 bytes, little endian) */
    fwrite(data, row_bytes, n, f);
    return fclose(f) == 0 ? 0 : -1;
}

int orc_load_desc(const char* path, uint64_t* n, size_t row_bytes, void* data, uint64_t cap)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t c;
    if (fread(&c, 8, 1, f) != 1) { fclose(f); return -3; }
    *n = c;
    if (data) {
        const uint64_t r = c < cap ? c : cap;
        if (fread(data, row_bytes, r, f) != r) { fclose(f); return -3; }
    }
    fclose(f);
    return 0;
}
