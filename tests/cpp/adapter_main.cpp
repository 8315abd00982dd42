// Host program in the reference's own language driving the two C++ faces of the boundary:
//   argv[1] = "knn"   : ArrayMatcher_r3dm<float>::Build + SearchNeighbours(NN=2) on a .desc pair,
//                       writes "q i0 d0 i1 d1" lines (what RegionsMatcherT::MatchDistanceRatio consumes)
//   argv[1] = "stage" : R3DComputeMatches::computeMatches on a matches directory
// Used by tests/test_gpu_cpp_host.py; also the compile check of include/*.hpp on CPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "r3d_compute_matches.hpp"
#include "regard3d_features.hpp"
#include "r3dm_array_matcher.hpp"

static bool read_desc(const char* path, int dim, std::vector<float>& out, int& n)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint64_t cnt = 0;
    bool ok = fread(&cnt, 8, 1, f) == 1;
    out.resize((size_t)cnt * dim);
    ok = ok && (cnt == 0 || fread(out.data(), sizeof(float) * dim, cnt, f) == cnt);
    fclose(f);
    n = (int)cnt;
    return ok;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "knn") && argc == 6) {
        const int dim = atoi(argv[4]);
        std::vector<float> a, b; int na = 0, nb = 0;
        if (!read_desc(argv[2], dim, a, na) || !read_desc(argv[3], dim, b, nb)) return 3;
        r3d_amd::ArrayMatcher_r3dm<float> matcher(0);
        if (!matcher.Build(a.data(), na, dim)) return 4;
        r3d_amd::IndMatches idx; std::vector<float> dist;
        if (!matcher.SearchNeighbours(b.data(), nb, &idx, &dist, 2)) return 5;
        FILE* o = fopen(argv[5], "w");
        for (int q = 0; q < nb; ++q)
            fprintf(o, "%u %u %.9g %u %.9g\n", idx[2 * q].i_, idx[2 * q].j_, dist[2 * q], idx[2 * q + 1].j_, dist[2 * q + 1]);
        fclose(o);
        int one_idx = -1; float one_d = 0;
        if (!matcher.SearchNeighbour(b.data(), &one_idx, &one_d) || one_idx != (int)idx[0].j_) return 6;
        return 0;
    }
    if (!strcmp(argv[1], "loop") && argc == 7) {
        // loop <I.desc> <J.desc> <dim> <n_searches> <out.txt>: the reference's use of a plugin (src/R3DComputeMatches.cpp:462-479):
        // Build once, then SearchNeighbours from an OpenMP loop.  Prints "<datasets+queries staged> <contexts used> <all equal>".
        const int dim = atoi(argv[4]), reps = atoi(argv[5]);
        std::vector<float> a, b; int na = 0, nb = 0;
        if (!read_desc(argv[2], dim, a, na) || !read_desc(argv[3], dim, b, nb)) return 3;
        r3d_amd::ArrayMatcher_r3dm<float> matcher(0);
        const uint64_t staged0 = matcher.viewsStaged();
        if (!matcher.Build(a.data(), na, dim)) return 4;
        std::vector<r3d_amd::IndMatches> idx(reps); std::vector<std::vector<float>> dist(reps);
        int failed = 0;
#pragma omp parallel for schedule(dynamic) num_threads(8)
        for (int r = 0; r < reps; ++r)
            if (!matcher.SearchNeighbours(b.data(), nb, &idx[r], &dist[r], 2)) {
#pragma omp atomic
                ++failed;
            }
        if (failed) return 5;
        bool same = true;
        for (int r = 1; r < reps; ++r)
            for (size_t k = 0; k < idx[0].size(); ++k)
                same = same && idx[r][k].j_ == idx[0][k].j_ && dist[r][k] == dist[0][k];
        FILE* o = fopen(argv[6], "w");
        for (int q = 0; q < nb; ++q)
            fprintf(o, "%u %u %.9g %u %.9g\n", idx[0][2 * q].i_, idx[0][2 * q].j_, dist[0][2 * q], idx[0][2 * q + 1].j_, dist[0][2 * q + 1]);
        fclose(o);
        printf("%llu %d %d\n", (unsigned long long)(matcher.viewsStaged() - staged0), matcher.contextsInUse(), same ? 1 : 0);
        return 0;
    }
    if (!strcmp(argv[1], "features") && argc >= 6) {
        // features <raw float32 gray image> <width> <height> <out.txt>: Regard3DFeatures::detectAndExtract on the GPU
        const uint32_t w = (uint32_t)atoi(argv[3]), h = (uint32_t)atoi(argv[4]);
        std::vector<float> img((size_t)w * h);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(img.data(), 4, img.size(), f) != img.size()) return 3;
        fclose(f);
        r3dm_ctx* ctx = nullptr;
        if (r3dm_create(0, &ctx) != R3DM_OK) { fprintf(stderr, "no gfx950 device\n"); return 7; }
        r3d_amd::FeatsR3D feats; r3d_amd::DescsR3D descs;
        r3d_amd::R3DFParams params;
        const bool ok = r3d_amd::Regard3DFeatures::detectAndExtract(ctx, {img.data(), w, h}, feats, descs, params);
        if (!ok) { fprintf(stderr, "detectAndExtract failed: %s\n", r3dm_last_error(ctx)); r3dm_destroy(ctx); return 7; }
        FILE* o = fopen(argv[5], "w");
        for (size_t k = 0; k < feats.size(); ++k) {
            fprintf(o, "%.9g %.9g %.9g %.9g", feats[k].x, feats[k].y, feats[k].scale, feats[k].orientation);
            for (float v : descs[k]) fprintf(o, " %.9g", v);
            fprintf(o, "\n");
        }
        fclose(o);
        r3dm_destroy(ctx);
        return 0;
    }
    if (!strcmp(argv[1], "stage") && argc >= 5) {
        // stage <matches_dir> <dim> <basename...>;  R3DM_TEST_DEVICES=N: the device-list constructor with N entries of device 0
        const char* ndev_env = getenv("R3DM_TEST_DEVICES");
        const std::vector<int> devices((size_t)(ndev_env ? atoi(ndev_env) : 1), 0);
        r3d_amd::R3DComputeMatches stage(devices);
        std::vector<r3d_amd::View> views;
        // synthetic views (regard3d_amd/synth.py): 4000 x 3000, f = 1.2 * width, principal point at the centre
        for (int k = 4; k < argc; ++k) views.push_back({(uint32_t)(k - 4), 4000, 3000, argv[k], 4800.0, 2000.0, 1500.0});
        stage.addViews(views);
        stage.setRegionsType(R3DM_F32, (uint32_t)atoi(argv[3]));
        stage.setProgressCallback([](float p, const char* msg, void*) { fprintf(stderr, "progress %.2f %s\n", p, msg); }, nullptr);
        r3d_amd::R3DFParams params;
        r3d_amd::R3DProjectPaths paths;
        paths.relativeMatchesPath_ = argv[2];
        // R3DM_TEST_ALGO selects the dispatch arm (default 9 = GPU brute force; 1..3 = KGraph presets)
        const char* algo_env = getenv("R3DM_TEST_ALGO");
        const int algo = algo_env ? atoi(algo_env) : r3d_amd::R3DComputeMatches::kMatchingAlgorithmGPU;
        if (getenv("R3DM_TEST_F32_TILES")) stage.setExactFastPaths(false);          // default: on
        // R3DM_TEST_ARMS=requested: approximate arms always on the graph matcher (default: on whichever matcher is faster for the views)
        const char* arms_env = getenv("R3DM_TEST_ARMS");
        if (arms_env && !strcmp(arms_env, "requested")) stage.setApproximateArmsPolicy(r3d_amd::R3DComputeMatches::kArmsAsRequested);
        const bool ok = stage.computeMatches(params, true, paths, 1, algo);
        if (!ok) { fprintf(stderr, "computeMatches failed: %s\n", stage.errorMessage().c_str()); return 7; }
        fprintf(stderr, "matcher %s\n", stage.lastMatchWasExhaustive() ? "exhaustive" : (stage.lastMatchWasHnsw() ? "hnsw" : "graph"));
        printf("%zu %zu\n", stage.getStatistics().putativeMatches_.size(), stage.getStatistics().fundamentalMatches_.size());
        return 0;
    }
    return 2;
}
