// Host program in the reference's own language driving the two C++ faces of the boundary:
//   argv[1] = "knn"   : ArrayMatcher_r3dm<float>::Build + SearchNeighbours(NN=2) on a .desc pair,
//                       writes "q i0 d0 i1 d1" lines (what RegionsMatcherT::MatchDistanceRatio consumes)
//   argv[1] = "stage" : R3DComputeMatches::computeMatches on a matches directory
// Used by tests/test_gpu_cpp_host.py; also the compile check of include/*.hpp on CPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
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
        // features <raw float32 gray image> <width> <height> <out.txt> [n_threads]: the reference's use of the feature face
        // (src/threads/R3DFeaturesThread.cpp:58-77,123-210): initAKAZESemaphore(1), then the static, handle-free
        // Regard3DFeatures::detectAndExtract(img, feats, descs, params) from n_threads (default CPUs + 1, at most 9) worker
        // threads at once, uninitializeAKAZESemaphore().  Thread 0's result goes to out.txt; prints
        // "<threads> <all results equal> <max detector calls in flight>".
        using r3d_amd::Regard3DFeatures;
        const uint32_t w = (uint32_t)atoi(argv[3]), h = (uint32_t)atoi(argv[4]);
        std::vector<float> img((size_t)w * h);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(img.data(), 4, img.size(), f) != img.size()) return 3;
        fclose(f);
        int nthreads = argc >= 7 ? atoi(argv[6]) : (int)std::min(9u, std::thread::hardware_concurrency() + 1);
        if (nthreads < 1) nthreads = 1;
        const std::vector<std::string> listed = Regard3DFeatures::getKeypointDetectors();
        if (listed.size() != 6 || listed[1] != "Fast-AKAZE" || Regard3DFeatures::getFeatureExtractors() != std::vector<std::string>{"LIOP"}) return 8;
        if (!Regard3DFeatures::initAKAZESemaphore(1)) return 8;
        Regard3DFeatures::resetMaxDetectorsInFlight();
        std::vector<Regard3DFeatures::FeatsR3D> feats(nthreads); std::vector<Regard3DFeatures::DescsR3D> descs(nthreads);
        std::vector<std::string> errors(nthreads);
        std::vector<std::thread> workers;
        for (int t = 0; t < nthreads; ++t)
            workers.emplace_back([&, t] {
                try {
                    const r3d_amd::ImageViewF view(img.data(), w, h);
                    Regard3DFeatures::R3DFParams params;          // keypointDetectorList_ = {"Fast-AKAZE"}, threshold_ = 0.001
                    Regard3DFeatures::detectAndExtract(view, feats[t], descs[t], params);
                } catch (const std::exception& e) { errors[t] = e.what(); }
            });
        for (std::thread& th : workers) th.join();
        const int in_flight = Regard3DFeatures::maxDetectorsInFlight();
        Regard3DFeatures::uninitializeAKAZESemaphore();
        for (const std::string& e : errors) if (!e.empty()) { fprintf(stderr, "detectAndExtract failed: %s\n", e.c_str()); return 7; }
        bool same = true;
        for (int t = 1; t < nthreads; ++t)
            same = same && feats[t].size() == feats[0].size() && descs[t] == descs[0] &&
                   !memcmp(feats[t].data(), feats[0].data(), feats[0].size() * sizeof(r3d_amd::FeatureR3D));
        // an unserved detector of the listed ones fails loudly, it never returns an empty result
        bool threw = false;
        try {
            Regard3DFeatures::R3DFParams params; params.keypointDetectorList_ = {"MSER"};
            Regard3DFeatures::FeatsR3D f2; Regard3DFeatures::DescsR3D d2;
            Regard3DFeatures::detectAndExtract(r3d_amd::ImageViewF(img.data(), w, h), f2, d2, params);
        } catch (const std::runtime_error&) { threw = true; }
        if (!threw) return 9;
        FILE* o = fopen(argv[5], "w");
        for (size_t k = 0; k < feats[0].size(); ++k) {
            fprintf(o, "%.9g %.9g %.9g %.9g", feats[0][k].x, feats[0][k].y, feats[0][k].scale, feats[0][k].orientation);
            for (float v : descs[0][k]) fprintf(o, " %.9g", v);
            fprintf(o, "\n");
        }
        fclose(o);
        printf("%d %d %d\n", nthreads, same ? 1 : 0, in_flight);
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
