// api_comm.cpp -- the one collective of the path, inside the product library: every rank ends up with the pairwise match graph of
// the whole collection.  SURVEY.md section 8e / BASELINE north star: "image-pairs shard embarrassingly across the 8 GPUs of one node
// with a single RCCL all-gather over xGMI to reassemble the pairwise match graph".  The reference's counterpart is the std::map that
// its OpenMP threads insert into under `omp critical` (/root/reference/src/R3DComputeMatches.cpp:465,481-487).
//
// Wire format of a rank's graphs (uint32 words): [n_graphs, len_0 .. len_{n-1}] then per graph [P, M_lo, M_hi, pairs (2 P),
// counts (P), matches (2 M)] -- what regard3d_amd/dist.py ships through torch.distributed; r3dm_graphs_pack / r3dm_graphs_unpack_merge
// are that format for ANY transport (MPI, files, torch), r3dm_allgather_graphs runs it over RCCL: one ncclAllGather of the sizes
// (8 bytes per rank), one of the payload padded to the largest rank.  RCCL is bound at run time (dlopen librccl.so): a host without
// it can still load the library, the comm entry points then report R3DM_ERR_UNSUPPORTED.  There is no fallback transport in here.
#include "r3dm_ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

extern "C" {
int  r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);
void r3dm_graph_free(r3dm_graph* g);
}

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl* r = [] {
        Rccl* x = new Rccl();
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            x->so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x->so) break;
        }
        if (!x->so) return x;
        x->GetUniqueId = reinterpret_cast<decltype(x->GetUniqueId)>(dlsym(x->so, "ncclGetUniqueId"));
        x->CommInitRank = reinterpret_cast<decltype(x->CommInitRank)>(dlsym(x->so, "ncclCommInitRank"));
        x->CommDestroy = reinterpret_cast<decltype(x->CommDestroy)>(dlsym(x->so, "ncclCommDestroy"));
        x->AllGather = reinterpret_cast<decltype(x->AllGather)>(dlsym(x->so, "ncclAllGather"));
        x->GetErrorString = reinterpret_cast<decltype(x->GetErrorString)>(dlsym(x->so, "ncclGetErrorString"));
        x->ok = x->GetUniqueId && x->CommInitRank && x->CommDestroy && x->AllGather && x->GetErrorString;
        return x;
    }();
    return *r;
}

void pack_graph(const r3dm_graph& g, std::vector<uint32_t>& out)
{
    const uint64_t P = g.pairs.size() / 2, M = g.matches.size();
    out.push_back((uint32_t)P); out.push_back((uint32_t)(M & 0xFFFFFFFFull)); out.push_back((uint32_t)(M >> 32));
    out.insert(out.end(), g.pairs.begin(), g.pairs.end());
    for (uint64_t p = 0; p < P; ++p) out.push_back((uint32_t)(g.offsets[p + 1] - g.offsets[p]));
    const size_t at = out.size();
    out.resize(at + 2 * M);
    if (M) memcpy(out.data() + at, g.matches.data(), 8 * M);
}

// words [at, at + len) of buf -> a graph; false on a malformed buffer
bool unpack_graph(const uint32_t* buf, uint64_t len, r3dm_graph& g)
{
    if (len < 3) return false;
    const uint64_t P = buf[0], M = (uint64_t)buf[1] | ((uint64_t)buf[2] << 32);
    if (len != 3 + 3 * P + 2 * M) return false;
    g.pairs.assign(buf + 3, buf + 3 + 2 * P);
    g.offsets.assign(1, 0);
    uint64_t run = 0;
    for (uint64_t p = 0; p < P; ++p) { run += buf[3 + 2 * P + p]; g.offsets.push_back(run); }
    if (run != M) return false;
    g.matches.resize(M);
    if (M) memcpy(g.matches.data(), buf + 3 + 3 * P, 8 * M);
    return true;
}

}  // namespace

struct r3dm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    DevBuf d_send, d_recv, d_sizes;
    PinBuf h_send, h_recv;
    std::string err;
    uint32_t last_device_graphs = 0;      // graphs of the last exchange whose payload was sent from their device mirror
};

extern "C" int r3dm_graphs_pack(const r3dm_graph* const* local, uint32_t n_graphs, uint32_t** words_out, uint64_t* n_words_out)
{
    if (!words_out || !n_words_out || (n_graphs && !local)) return R3DM_ERR_INVALID;
    *words_out = nullptr; *n_words_out = 0;
    try {
        std::vector<uint32_t> buf(1 + (size_t)n_graphs, 0u);
        buf[0] = n_graphs;
        for (uint32_t k = 0; k < n_graphs; ++k) {
            if (!local[k]) return R3DM_ERR_INVALID;
            if (local[k]->pairs.size() / 2 > 0xFFFFFFFFull) return R3DM_ERR_UNSUPPORTED;     // (the pair count travels as one word)
            const size_t at = buf.size();
            pack_graph(*local[k], buf);
            if (buf.size() - at > 0xFFFFFFFFull) return R3DM_ERR_UNSUPPORTED;
            buf[1 + k] = (uint32_t)(buf.size() - at);
        }
        uint32_t* out = static_cast<uint32_t*>(malloc(std::max<size_t>(buf.size(), 1) * 4));
        if (!out) return R3DM_ERR_NOMEM;
        memcpy(out, buf.data(), buf.size() * 4);
        *words_out = out; *n_words_out = buf.size();
        return R3DM_OK;
    } catch (...) { return R3DM_ERR_NOMEM; }
}

extern "C" void r3dm_words_free(uint32_t* words) { free(words); }

extern "C" int r3dm_graphs_unpack_merge(const uint32_t* const* rank_words, const uint64_t* rank_n_words, uint32_t world, uint32_t n_graphs,
                                        r3dm_graph** merged_out)
{
    if (!rank_words || !rank_n_words || !merged_out || world == 0) return R3DM_ERR_INVALID;
    for (uint32_t k = 0; k < n_graphs; ++k) merged_out[k] = nullptr;
    try {
        std::vector<std::vector<std::unique_ptr<r3dm_graph>>> parts(n_graphs);
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t* b = rank_words[r];
            const uint64_t n = rank_n_words[r];
            if (!b || n < 1 || b[0] != n_graphs || n < 1 + (uint64_t)n_graphs) return R3DM_ERR_INVALID;
            uint64_t at = 1 + (uint64_t)n_graphs;
            for (uint32_t k = 0; k < n_graphs; ++k) {
                const uint64_t len = b[1 + k];
                if (at + len > n) return R3DM_ERR_INVALID;
                auto g = std::unique_ptr<r3dm_graph>(new r3dm_graph());
                if (!unpack_graph(b + at, len, *g)) return R3DM_ERR_INVALID;
                parts[k].push_back(std::move(g));
                at += len;
            }
            if (at != n) return R3DM_ERR_INVALID;                   // (words behind the last graph: not a buffer of this format)
        }
        for (uint32_t k = 0; k < n_graphs; ++k) {
            std::vector<const r3dm_graph*> ptrs;
            for (auto& g : parts[k]) ptrs.push_back(g.get());
            const int rc = r3dm_graph_merge(ptrs.data(), (uint32_t)ptrs.size(), &merged_out[k]);
            if (rc != R3DM_OK) {
                for (uint32_t j = 0; j < n_graphs; ++j) { if (merged_out[j]) r3dm_graph_free(merged_out[j]); merged_out[j] = nullptr; }
                return rc;
            }
        }
        return R3DM_OK;
    } catch (...) { return R3DM_ERR_NOMEM; }
}

extern "C" int r3dm_comm_unique_id(void* id_out_128)
{
    if (!id_out_128) return R3DM_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id around as 128 bytes");
    if (!rccl().ok) return R3DM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return R3DM_ERR_HIP;
    memcpy(id_out_128, &id, 128);
    return R3DM_OK;
}

extern "C" int r3dm_comm_create(const void* id_128, int rank, int world, int device_id, r3dm_comm** out)
{
    if (!out || !id_128 || world < 1 || rank < 0 || rank >= world) return R3DM_ERR_INVALID;
    *out = nullptr;
    if (!rccl().ok) return R3DM_ERR_UNSUPPORTED;
    if (hipSetDevice(device_id) != hipSuccess) return R3DM_ERR_NO_DEVICE;
    auto c = std::unique_ptr<r3dm_comm>(new (std::nothrow) r3dm_comm());
    if (!c) return R3DM_ERR_NOMEM;
    c->rank = rank; c->world = world; c->device = device_id;
    ncclUniqueId id;
    memcpy(&id, id_128, 128);
    if (rccl().CommInitRank(&c->comm, world, id, rank) != ncclSuccess) return R3DM_ERR_HIP;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rccl().CommDestroy(c->comm); return R3DM_ERR_HIP; }
    *out = c.release();
    return R3DM_OK;
}

extern "C" void r3dm_comm_destroy(r3dm_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    c->d_send.release(); c->d_recv.release(); c->d_sizes.release(); c->h_send.release(); c->h_recv.release();
    delete c;
}

extern "C" int r3dm_comm_rank(const r3dm_comm* c) { return c ? c->rank : -1; }
extern "C" int r3dm_comm_world(const r3dm_comm* c) { return c ? c->world : -1; }
extern "C" const char* r3dm_comm_last_error(const r3dm_comm* c) { return c ? c->err.c_str() : "null communicator"; }

#define CHIP(call)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e__); return R3DM_ERR_HIP; } \
    } while (0)
#define CNCCL(call)                                                                             \
    do {                                                                                        \
        ncclResult_t e__ = (call);                                                              \
        if (e__ != ncclSuccess) { c->err = std::string(#call) + ": " + rccl().GetErrorString(e__); return R3DM_ERR_HIP; } \
    } while (0)

extern "C" int r3dm_comm_last_device_graphs(const r3dm_comm* c) { return c ? (int)c->last_device_graphs : -1; }

// The exchange.  A rank takes part in every collective of the call whatever happened to it while it PREPARED its share: a rank that
// cannot pack its graphs or allocate the exchange buffers says so in the words it contributes (size ~0 / status 1), a rank whose
// copies into the send buffer fail poisons the first word of what it sends, and ALL ranks return an error together -- a rank that
// left early would leave the others blocked inside RCCL.  What is NOT covered (nothing a rank could contribute): hipSetDevice, the
// 16 (W + 1)-byte buffer of the size collective, and a failing collective or stream itself -- a device that is gone.
// A local graph with a device mirror on this communicator's GPU (r3dm_set_device_graphs; GraphDev) goes on the wire from device memory:
// three device-to-device copies behind a 12-byte header -- its payload never visits the host on the way out.  A graph without one is
// packed on the host and uploaded, as before.  The gathered payload comes back to the host either way: the merged graphs are host
// objects (PairWiseMatches feeds files and the SfM stage).
extern "C" int r3dm_allgather_graphs(r3dm_comm* c, const r3dm_graph* const* local, uint32_t n_graphs, r3dm_graph** merged_out)
{
    if (!c || !merged_out || (n_graphs && !local)) return R3DM_ERR_INVALID;
    for (uint32_t k = 0; k < n_graphs; ++k) merged_out[k] = nullptr;
    try {
        CHIP(hipSetDevice(c->device));
        const uint32_t W = (uint32_t)c->world;
        constexpr unsigned long long kFailed = ~0ull;
        // ---- what this rank sends: per graph the word count, and whether the device mirror serves it
        int local_rc = R3DM_OK;
        std::vector<uint64_t> len(n_graphs, 0);
        std::vector<unsigned char> on_dev(n_graphs, 0);
        unsigned long long n_words = 1ull + n_graphs;
        for (uint32_t k = 0; k < n_graphs && local_rc == R3DM_OK; ++k) {
            const r3dm_graph* g = local[k];
            if (!g) { local_rc = R3DM_ERR_INVALID; break; }
            const uint64_t P = g->pairs.size() / 2, M = g->matches.size();
            if (P > 0xFFFFFFFFull || 3 + 3 * P + 2 * M > 0xFFFFFFFFull) { local_rc = R3DM_ERR_UNSUPPORTED; break; }
            len[k] = 3 + 3 * P + 2 * M;
            on_dev[k] = g->dev.valid && g->dev.device == c->device && g->dev.P == P && g->dev.M == M;
            n_words += len[k];
        }
        // ---- sizes: 8 bytes per rank
        CHIP(c->d_sizes.ensure(8 * (size_t)(2 * W + 2)));
        unsigned long long* d_all = c->d_sizes.as<unsigned long long>();        // [W] gathered sizes, [W] gathered status, then mine x 2
        const unsigned long long mine = local_rc == R3DM_OK ? n_words : kFailed;
        CHIP(hipMemcpyAsync(d_all + 2 * W, &mine, 8, hipMemcpyHostToDevice, c->stream));
        CNCCL(rccl().AllGather(d_all + 2 * W, d_all, 1, ncclUint64, c->comm, c->stream));
        std::vector<unsigned long long> sizes(W);
        CHIP(hipMemcpyAsync(sizes.data(), d_all, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
        CHIP(hipStreamSynchronize(c->stream));
        unsigned long long mx = 1;
        for (uint32_t r = 0; r < W; ++r) {
            if (sizes[r] == kFailed) { c->err = "r3dm_allgather_graphs: rank " + std::to_string(r) + " could not pack its graphs (null graph, or more than 2^32 words in one)"; return local_rc != R3DM_OK ? local_rc : R3DM_ERR_INVALID; }
            mx = std::max(mx, sizes[r]);
        }
        // ---- buffers, then a status word per rank: nobody enters the payload collective unless everybody can
        unsigned long long status = 0;
        if (c->d_send.ensure(4 * (size_t)mx) != hipSuccess || c->d_recv.ensure(4 * (size_t)mx * W) != hipSuccess ||
            c->h_send.ensure(4 * (size_t)mx) != hipSuccess || c->h_recv.ensure(4 * (size_t)mx * W) != hipSuccess) { status = 1; (void)hipGetLastError(); }
        CHIP(hipMemcpyAsync(d_all + 2 * W + 1, &status, 8, hipMemcpyHostToDevice, c->stream));
        CNCCL(rccl().AllGather(d_all + 2 * W + 1, d_all + W, 1, ncclUint64, c->comm, c->stream));
        std::vector<unsigned long long> stat(W);
        CHIP(hipMemcpyAsync(stat.data(), d_all + W, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
        CHIP(hipStreamSynchronize(c->stream));
        for (uint32_t r = 0; r < W; ++r)
            if (stat[r]) { c->err = "r3dm_allgather_graphs: rank " + std::to_string(r) + " is out of memory for the exchange buffers"; return R3DM_ERR_NOMEM; }
        // ---- my words in d_send: the host writes headers (and whole graphs without a mirror) into the pinned image, uploads them in
        // runs, and the mirrors fill their places device to device
        // (from here to the payload collective a failed copy is remembered, not returned: the collective is entered either way)
        hipError_t late = hipSuccess; const char* late_what = "";
#define CSOFT(call) do { if (late == hipSuccess) { late = (call); if (late != hipSuccess) late_what = #call; } } while (0)
        uint32_t* hs = static_cast<uint32_t*>(c->h_send.p);
        uint32_t* ds = c->d_send.as<uint32_t>();
        hs[0] = n_graphs;
        for (uint32_t k = 0; k < n_graphs; ++k) hs[1 + k] = (uint32_t)len[k];
        size_t at = 1 + (size_t)n_graphs, run0 = 0;                              // [run0, at) = host words not yet uploaded
        c->last_device_graphs = 0;
        for (uint32_t k = 0; k < n_graphs; ++k) {
            const r3dm_graph& g = *local[k];
            const uint64_t P = g.pairs.size() / 2, M = g.matches.size();
            hs[at] = (uint32_t)P; hs[at + 1] = (uint32_t)(M & 0xFFFFFFFFull); hs[at + 2] = (uint32_t)(M >> 32);
            if (on_dev[k]) {
                CSOFT(hipMemcpyAsync(ds + run0, hs + run0, 4 * (at + 3 - run0), hipMemcpyHostToDevice, c->stream));
                uint32_t* d = ds + at + 3;
                if (P) CSOFT(hipMemcpyAsync(d, g.dev.pairs.p, 8 * (size_t)P, hipMemcpyDeviceToDevice, c->stream));
                if (P) CSOFT(hipMemcpyAsync(d + 2 * P, g.dev.counts.p, 4 * (size_t)P, hipMemcpyDeviceToDevice, c->stream));
                if (M) CSOFT(hipMemcpyAsync(d + 3 * P, g.dev.matches.p, 8 * (size_t)M, hipMemcpyDeviceToDevice, c->stream));
                at += (size_t)len[k];
                run0 = at;
                c->last_device_graphs += 1;
            } else {
                uint32_t* o = hs + at + 3;
                if (P) memcpy(o, g.pairs.data(), 8 * (size_t)P);
                for (uint64_t p = 0; p < P; ++p) o[2 * P + p] = (uint32_t)(g.offsets[p + 1] - g.offsets[p]);
                if (M) memcpy(o + 3 * P, g.matches.data(), 8 * (size_t)M);
                at += (size_t)len[k];
            }
        }
        if (mx > at) memset(hs + at, 0, 4 * (size_t)(mx - at));                   // padding up to the largest rank
        if (mx > run0) CSOFT(hipMemcpyAsync(ds + run0, hs + run0, 4 * (size_t)(mx - run0), hipMemcpyHostToDevice, c->stream));
        // ---- payload, padded to the largest rank: the one exchange (RCCL over xGMI)
        if (late != hipSuccess) {
            // my send buffer is not what its header says: poison its first word (the unpacker checks it against n_graphs on every rank)
            static const uint32_t kPoison = 0xFFFFFFFFu;
            (void)hipGetLastError();
            (void)hipMemcpyAsync(ds, &kPoison, 4, hipMemcpyHostToDevice, c->stream);
        }
#undef CSOFT
        CNCCL(rccl().AllGather(c->d_send.p, c->d_recv.p, (size_t)mx, ncclUint32, c->comm, c->stream));
        if (late != hipSuccess) { (void)hipStreamSynchronize(c->stream); c->err = std::string(late_what) + ": " + hipGetErrorString(late); return R3DM_ERR_HIP; }
        CHIP(hipMemcpyAsync(c->h_recv.p, c->d_recv.p, 4 * (size_t)mx * W, hipMemcpyDeviceToHost, c->stream));
        CHIP(hipStreamSynchronize(c->stream));
        std::vector<const uint32_t*> ptrs(W);
        std::vector<uint64_t> lens(W);
        for (uint32_t r = 0; r < W; ++r) { ptrs[r] = static_cast<const uint32_t*>(c->h_recv.p) + (size_t)r * mx; lens[r] = sizes[r]; }
        const int rc = r3dm_graphs_unpack_merge(ptrs.data(), lens.data(), W, n_graphs, merged_out);
        if (rc != R3DM_OK) c->err = "r3dm_allgather_graphs: a rank sent a malformed buffer";
        return rc;
    } catch (...) { c->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}
