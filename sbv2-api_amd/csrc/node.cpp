// Multi-GPU entry points of libsbv2_hip.so (the reference has nothing here: it is one process, one device, batch 1).
//
// Utterances are independent, so a batch is SHARDED: sorted by cost, dealt to the GPUs (longest-processing-time-first), every GPU
// holds a full weight replica and runs DeBERTa + VITS + HiFi-GAN for its shard; the only exchange is the gather of PCM to rank 0:
// one ncclAllGather of the per-rank sample counts + grouped ncclSend / ncclRecv of the f32 samples (RCCL over xGMI: every peer has its
// own link into rank 0, there is no ring and no reduction anywhere on the path).
//
// Two shapes of the same thing:
//   sbv2_comm_*  one PROCESS per GPU (how bench.py is launched: `torchrun`-style RANK / WORLD_SIZE env, no torch inside): the ranks share an
//                ncclUniqueId, each drives its own pipeline and calls sbv2_comm_gather_pcm.
//   sbv2_node_*  one process, N devices (how a server behind sbv2_api would use a node: one handle, `sbv2_node_synthesize` = the
//                `sbv2_synthesize_batch(ctx, ...)` of SURVEY.md §8b): one host thread + stream set per device, ncclCommInitAll.
//
// RCCL is loaded with dlopen("librccl.so.1") the first time a communicator is needed: the library has no link-time dependency on it and
// single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <numeric>
#include <thread>

#include "api_internal.h"

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
        if (!r.h) {
            err = std::string("RCCL is not available (dlopen librccl.so.1: ") + dlerror() + ")";
            return;
        }
#define SYM(f)                                                       \
    r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, "nccl" #f));    \
    if (!r.f) err = "librccl.so.1 does not export nccl" #f;
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(AllReduce) SYM(AllGather) SYM(Send) SYM(Recv) SYM(GroupStart)
        SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
    });
    if (!err.empty()) throw Error(err);
    return r;
}

#define NCCL_CHECK(expr)                                                                                                     \
    do {                                                                                                                     \
        ncclResult_t _r = (expr);                                                                                            \
        if (_r != ncclSuccess) throw Error(std::string(#expr) + ": " + rccl().GetErrorString(_r) + " (node.cpp:" + std::to_string(__LINE__) + ")"); \
    } while (0)

// device buffer that only grows
struct GrowBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* get(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
            HIP_CHECK(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    ~GrowBuf() {
        if (p) (void)hipFree(p);
    }
};

// Longest-processing-time-first: utterances by descending cost (ties by index), each to the least-loaded rank (ties by rank).  Loads differ by
// at most one utterance's cost; the COUNT per rank is not bounded by ceil(n / world).  Same rule as sbv2-api_amd/shard.py::deal.
std::vector<int> deal(const std::vector<int64_t>& costs, int world) {
    const int n = (int)costs.size();
    std::vector<int> order(n), rank_of(n, 0);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return costs[a] > costs[b]; });
    std::vector<int64_t> load(world, 0);
    for (int i : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        rank_of[i] = best;
        load[best] += costs[i];
    }
    return rank_of;
}

// The gather of one dealt batch: rank r's message = the PCM of its utterances in ascending caller index, messages in rank order in the staging
// buffer; table[3 e] = {offset in the staging buffer, offset in the caller's utterance order, samples} for every utterance (k_copy_segments).
void gather_plan(int n, const int64_t* pcm_lens, const int* rank_of, int world, std::vector<int64_t>& counts, std::vector<int64_t>& table) {
    counts.assign(world, 0);
    table.assign((size_t)3 * n, 0);
    std::vector<int64_t> out_off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        SBV2_REQUIRE(rank_of[i] >= 0 && rank_of[i] < world && pcm_lens[i] >= 0, "gather plan: bad rank or length");
        counts[rank_of[i]] += pcm_lens[i];
        out_off[i + 1] = out_off[i] + pcm_lens[i];
    }
    int e = 0;
    int64_t so = 0;
    for (int r = 0; r < world; ++r)
        for (int i = 0; i < n; ++i)
            if (rank_of[i] == r) {
                table[3 * e] = so;
                table[3 * e + 1] = out_off[i];
                table[3 * e + 2] = pcm_lens[i];
                so += pcm_lens[i];
                ++e;
            }
}

}  // namespace

struct sbv2_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int64_t* d_counts = nullptr;   // [world + 1]: [0, world) gathered, [world] = this rank's value
    double* d_val = nullptr;       // [2]
    GrowBuf stage;                 // root: every rank's PCM, rank order
    hipEvent_t ev = nullptr;
    hipStream_t copy = nullptr;    // root: device -> host copies, overlapped with the receives still in flight
    std::vector<hipEvent_t> gev;   // root: one event per receive group
};

struct sbv2_node {
    struct Dev {
        int device = 0;
        std::unique_ptr<sbv2_bert> bert;
        std::unique_ptr<sbv2_vits> vits;
        hipStream_t xfer = nullptr;   // gather stream of this device
        hipEvent_t ev = nullptr;
        ncclComm_t comm = nullptr;
    };
    std::vector<Dev> devs;
    bool use_rccl = false;
    GrowBuf stage, ordered, table;    // on device 0
    std::vector<int32_t> last_rank_of;
};

extern "C" {

int sbv2_deal(int64_t n, const int64_t* costs, int world, int32_t* rank_of) {
    API_BEGIN
    SBV2_REQUIRE(n >= 0 && world >= 1 && (n == 0 || (costs && rank_of)), "bad arguments");
    const std::vector<int> r = deal(std::vector<int64_t>(costs, costs + n), world);
    for (int64_t i = 0; i < n; ++i) rank_of[i] = r[i];
    API_END
}

int sbv2_gather_plan(int64_t n, const int64_t* pcm_lens, const int32_t* rank_of, int world, int64_t* counts, int64_t* table) {
    API_BEGIN
    SBV2_REQUIRE(n >= 0 && n <= 0x7FFFFFFF && world >= 1 && counts && (n == 0 || (pcm_lens && rank_of && table)), "bad arguments");
    std::vector<int> ro(rank_of, rank_of + n);
    std::vector<int64_t> c, t;
    gather_plan((int)n, pcm_lens, ro.data(), world, c, t);
    std::copy(c.begin(), c.end(), counts);
    std::copy(t.begin(), t.end(), table);
    API_END
}

// ---- one process per GPU ------------------------------------------------------------------------------------------------------------
int sbv2_comm_unique_id(uint8_t* id128) {
    API_BEGIN
    SBV2_REQUIRE(id128, "bad arguments");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_CHECK(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, 128);
    API_END
}

int sbv2_comm_create(const uint8_t* id128, int rank, int world, int device, sbv2_comm** out) {
    API_BEGIN
    SBV2_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<sbv2_comm> c(new sbv2_comm);
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    NCCL_CHECK(rccl().CommInitRank(&c->comm, world, id, rank));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    c->gev.resize(world + 1);
    for (auto& e : c->gev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c->d_counts), sizeof(int64_t) * (world + 1)));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&c->d_val), sizeof(double) * 2));
    *out = c.release();
    API_END
}

void sbv2_comm_destroy(sbv2_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    try {
        if (c->comm) (void)rccl().CommDestroy(c->comm);
    } catch (...) {
    }
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_val) (void)hipFree(c->d_val);
    if (c->ev) (void)hipEventDestroy(c->ev);
    for (auto e : c->gev)
        if (e) (void)hipEventDestroy(e);
    if (c->copy) {
        (void)hipStreamSynchronize(c->copy);
        (void)hipStreamDestroy(c->copy);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sbv2_comm_rank(const sbv2_comm* c) { return c ? c->rank : -1; }
int sbv2_comm_world(const sbv2_comm* c) { return c ? c->world : -1; }

// max over ranks, in place (also the barrier: every rank leaves after every rank has entered)
int sbv2_comm_max_f64(sbv2_comm* c, double* v) {
    API_BEGIN
    SBV2_REQUIRE(c && v, "bad arguments");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipMemcpyAsync(c->d_val, v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCL_CHECK(rccl().AllReduce(c->d_val, c->d_val + 1, 1, ncclDouble, ncclMax, c->comm, c->stream));
    HIP_CHECK(hipMemcpyAsync(v, c->d_val + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    API_END
}
int sbv2_comm_barrier(sbv2_comm* c) {
    double z = 0;
    return sbv2_comm_max_f64(c, &z);
}

// Gather of the (variable-length) PCM of one pipeline run per rank to `root`: counts[world] (samples per rank) are filled on EVERY rank; on
// the root dst_host receives the ranks' concatenated PCM in rank order (capacity = samples it can hold; a longer result is refused after
// the exchange has completed, so that no rank is left waiting).  dst_host may be pinned memory (sbv2_host_alloc).
int sbv2_comm_gather_pcm(sbv2_comm* c, sbv2_pipeline* p, int64_t ticket, int root, float* dst_host, int64_t capacity, int64_t* counts) {
    API_BEGIN
    SBV2_REQUIRE(c && p && counts && root >= 0 && root < c->world, "bad arguments");
    SBV2_REQUIRE(c->rank != root || dst_host, "the root needs a destination buffer");
    VitsModel& vm = p->vm(p->ctx_of(ticket));
    SBV2_REQUIRE(vm.device() == c->device, "pipeline and communicator live on different devices");
    TraceRange tr("gather_pcm");
    HIP_CHECK(hipSetDevice(c->device));
    Rccl& R = rccl();
    // the gather stream waits for the run's kernels (event, not a host wait)
    HIP_CHECK(hipEventRecord(c->ev, vm.stream()));
    HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev, 0));
    const int64_t mine = vm.pcm_total();
    HIP_CHECK(hipMemcpyAsync(c->d_counts + c->world, &mine, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    NCCL_CHECK(R.AllGather(c->d_counts + c->world, c->d_counts, 1, ncclInt64, c->comm, c->stream));
    HIP_CHECK(hipMemcpyAsync(counts, c->d_counts, sizeof(int64_t) * c->world, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    int64_t total = 0;
    for (int r = 0; r < c->world; ++r) {
        SBV2_REQUIRE(counts[r] >= 0, "negative PCM count received");
        total += counts[r];
    }
    if (c->rank == root) {
        // The serial part of an N-GPU step is this rank's device -> host traffic (N x 59 MB at batch 32 x U128 over one PCIe link), so it starts
        // as early as it can: the root's own block goes to the host straight from the run's buffer while the peers' blocks are still on
        // their xGMI links.  The peers' receives are ONE ncclGroup by default (the shape every RCCL user runs); SBV2_GATHER_GROUP=n posts them in groups
        // of n whose copies to the host overlap the next group's receives: that variant has never executed with a real peer (no >= 2-GPU box was
        // available to this build), so it stays opt-in until test_comm_two_ranks_gather_over_rccl has run it.
        float* st = static_cast<float*>(c->stage.get(sizeof(float) * (size_t)std::max<int64_t>(total, 1)));
        std::vector<int64_t> offs(c->world + 1, 0);
        for (int r = 0; r < c->world; ++r) offs[r + 1] = offs[r] + counts[r];
        const bool fits = total <= capacity;
        static const int per = getenv("SBV2_GATHER_GROUP") && atoi(getenv("SBV2_GATHER_GROUP")) > 0 ? atoi(getenv("SBV2_GATHER_GROUP")) : 1 << 20;
        HIP_CHECK(hipEventRecord(c->gev[c->world], c->stream));          // (the run has finished: c->stream waited for it above)
        HIP_CHECK(hipStreamWaitEvent(c->copy, c->gev[c->world], 0));
        if (fits && mine > 0)
            HIP_CHECK(hipMemcpyAsync(dst_host + offs[root], vm.pcm_device(), sizeof(float) * (size_t)mine, hipMemcpyDeviceToHost, c->copy));
        std::vector<int> peers;
        for (int r = 0; r < c->world; ++r)
            if (r != root && counts[r] > 0) peers.push_back(r);
        for (size_t g0 = 0, gi = 0; g0 < peers.size(); g0 += per, ++gi) {
            const size_t g1 = std::min(peers.size(), g0 + per);
            NCCL_CHECK(R.GroupStart());
            for (size_t k = g0; k < g1; ++k) NCCL_CHECK(R.Recv(st + offs[peers[k]], (size_t)counts[peers[k]], ncclFloat, peers[k], c->comm, c->stream));
            NCCL_CHECK(R.GroupEnd());
            HIP_CHECK(hipEventRecord(c->gev[gi], c->stream));
            HIP_CHECK(hipStreamWaitEvent(c->copy, c->gev[gi], 0));
            if (fits)
                for (size_t k = g0; k < g1; ++k)
                    HIP_CHECK(hipMemcpyAsync(dst_host + offs[peers[k]], st + offs[peers[k]], sizeof(float) * (size_t)counts[peers[k]],
                                             hipMemcpyDeviceToHost, c->copy));
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        HIP_CHECK(hipStreamSynchronize(c->copy));
        SBV2_REQUIRE(fits, "PCM buffer too small: " + std::to_string(capacity) + " < " + std::to_string(total));
    } else {
        if (mine > 0) {
            NCCL_CHECK(R.GroupStart());
            NCCL_CHECK(R.Send(vm.pcm_device(), (size_t)mine, ncclFloat, root, c->comm, c->stream));
            NCCL_CHECK(R.GroupEnd());
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));   // the PCM lives in the run's workspace: it may be reused once this returns
    }
    API_END
}

// ---- one process, N devices ---------------------------------------------------------------------------------------------------------
void sbv2_node_destroy(sbv2_node* nd) {
    if (!nd) return;
    for (auto& d : nd->devs) {
        (void)hipSetDevice(d.device);
        if (d.xfer) (void)hipStreamSynchronize(d.xfer);
        try {
            if (d.comm) (void)rccl().CommDestroy(d.comm);
        } catch (...) {
        }
        if (d.ev) (void)hipEventDestroy(d.ev);
        if (d.xfer) (void)hipStreamDestroy(d.xfer);
    }
    if (!nd->devs.empty()) (void)hipSetDevice(nd->devs[0].device);
    delete nd;
}

// devices[ndev]: HIP device ordinals; the same ordinal may appear more than once (several shards on one GPU: used by the tests on a
// one-GPU box; such peers exchange by device-to-device copies).  With all ordinals distinct and ndev > 1 the gather runs over RCCL
// (ncclCommInitAll) unless SBV2_NODE_GATHER=peer selects hipMemcpyPeerAsync.
int sbv2_node_create(const uint8_t* bert_model, size_t bert_len, const uint8_t* vits_model, size_t vits_len, const int* devices, int ndev,
                     sbv2_node** out) {
    sbv2_node* raw = nullptr;
    try {
        SBV2_REQUIRE(bert_model && vits_model && devices && ndev >= 1 && ndev <= 64 && out, "bad arguments");
        raw = new sbv2_node;
        Blob bb = load_model_bytes(bert_model, bert_len, 1), vb = load_model_bytes(vits_model, vits_len, 2);
        raw->devs.resize(ndev);
        bool distinct = true;
        for (int i = 0; i < ndev; ++i)
            for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
        // replicas are loaded concurrently, one host thread per device (weight packing is host work)
        std::vector<std::string> errs(ndev);
        std::vector<std::thread> th;
        for (int i = 0; i < ndev; ++i)
            th.emplace_back([&, i] {
                try {
                    sbv2_node::Dev& d = raw->devs[i];
                    d.device = devices[i];
                    HIP_CHECK(hipSetDevice(d.device));
                    d.bert.reset(new sbv2_bert);
                    d.bert->m.reset(new BertModel(bb, d.device));
                    d.vits.reset(new sbv2_vits);
                    d.vits->m.reset(new VitsModel(vb, d.device));
                    HIP_CHECK(hipStreamCreateWithFlags(&d.xfer, hipStreamNonBlocking));
                    HIP_CHECK(hipEventCreateWithFlags(&d.ev, hipEventDisableTiming));
                } catch (const std::exception& e) {
                    errs[i] = e.what();
                }
            });
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (!e.empty()) throw Error(e);
        const char* g = getenv("SBV2_NODE_GATHER");
        raw->use_rccl = distinct && ndev > 1 && !(g && std::string(g) == "peer");
        if (raw->use_rccl) {
            std::vector<ncclComm_t> comms(ndev);
            NCCL_CHECK(rccl().CommInitAll(comms.data(), ndev, devices));
            for (int i = 0; i < ndev; ++i) raw->devs[i].comm = comms[i];
        } else if (distinct && ndev > 1) {
            for (int i = 1; i < ndev; ++i) {
                HIP_CHECK(hipSetDevice(devices[0]));
                (void)hipDeviceEnablePeerAccess(devices[i], 0);   // already enabled is fine
                (void)hipGetLastError();
            }
        }
        SBV2_REQUIRE(raw->devs[0].bert->m->cfg().hidden == raw->devs[0].vits->m->cfg().bert_dim,
                     "DeBERTa hidden size does not match the VITS bert_proj input");
        *out = raw;
        return 0;
    } catch (const std::exception& e) {
        set_last_error(e.what());
    } catch (...) {
        set_last_error("unknown error");
    }
    sbv2_node_destroy(raw);
    return 1;
}

int sbv2_node_devices(const sbv2_node* nd) { return nd ? (int)nd->devs.size() : 0; }
int sbv2_node_uses_rccl(const sbv2_node* nd) { return nd && nd->use_rccl ? 1 : 0; }
// rank (index into `devices`) that synthesised utterance i of the last call
int sbv2_node_last_deal(const sbv2_node* nd, int32_t* rank_of, int64_t n) {
    API_BEGIN
    SBV2_REQUIRE(nd && rank_of && n == (int64_t)nd->last_rank_of.size(), "bad arguments");
    std::copy(nd->last_rank_of.begin(), nd->last_rank_of.end(), rank_of);
    API_END
}

// The whole hot path for a batch on all devices of the node (= `sbv2_synthesize_batch(ctx, ...)` of SURVEY.md §8b).  Same inputs as
// sbv2_pipeline_run; pcm_lens[n] and pcm_host (concatenated PCM in the caller's utterance order, capacity samples) are outputs.  Every
// utterance's PCM is what a single-GPU call of the whole batch returns for it (noise streams are keyed by the caller's utterance index).
int sbv2_node_synthesize(sbv2_node* nd, const sbv2_batch* batch, const int64_t* token_ids, const int64_t* s_lens, const int64_t* word2ph,
                         int64_t* pcm_lens, float* pcm_host, int64_t capacity) {
    API_BEGIN
    SBV2_REQUIRE(nd && token_ids && s_lens && word2ph && pcm_lens && pcm_host, "bad arguments");
    const VitsBatch full = to_batch(batch);
    const int n = full.n, ndev = (int)nd->devs.size();
    const int style_dim = nd->devs[0].vits->m->cfg().style_dim;
    // offsets of every utterance in the concatenated per-token / per-BERT-token arrays
    std::vector<int64_t> toff(n + 1, 0), soff(n + 1, 0), cost(n);
    for (int i = 0; i < n; ++i) {
        SBV2_REQUIRE(full.t_lens[i] >= 1 && s_lens[i] >= 1, "bad utterance length");
        toff[i + 1] = toff[i] + full.t_lens[i];
        soff[i + 1] = soff[i] + s_lens[i];
    }
    for (int i = 0; i < n; ++i) {
        // cost ~ frames: the decoder dominates; known exactly when durations are teacher-forced, else ~ text length
        int64_t c = 0;
        if (full.forced_durations)
            for (int64_t t = toff[i]; t < toff[i + 1]; ++t) c += full.forced_durations[t];
        else c = 3 * full.t_lens[i];
        cost[i] = std::max<int64_t>(c, 1);
    }
    const std::vector<int> rank_of = deal(cost, ndev);
    nd->last_rank_of.assign(rank_of.begin(), rank_of.end());

    struct Shard {
        std::vector<int64_t> ids, t_lens, x, tones, langs, sids, forced, tok, s_lens, w2p, lens;
        std::vector<float> styles;
        std::string err;
    };
    std::vector<Shard> sh(ndev);
    for (int i = 0; i < n; ++i) {
        Shard& s = sh[rank_of[i]];
        s.ids.push_back(i);
        s.t_lens.push_back(full.t_lens[i]);
        s.sids.push_back(full.sids[i]);
        s.s_lens.push_back(s_lens[i]);
        s.x.insert(s.x.end(), full.phones + toff[i], full.phones + toff[i + 1]);
        s.tones.insert(s.tones.end(), full.tones + toff[i], full.tones + toff[i + 1]);
        s.langs.insert(s.langs.end(), full.langs + toff[i], full.langs + toff[i + 1]);
        if (full.forced_durations) s.forced.insert(s.forced.end(), full.forced_durations + toff[i], full.forced_durations + toff[i + 1]);
        s.tok.insert(s.tok.end(), token_ids + soff[i], token_ids + soff[i + 1]);
        s.w2p.insert(s.w2p.end(), word2ph + soff[i], word2ph + soff[i + 1]);
        s.styles.insert(s.styles.end(), full.styles + (size_t)i * style_dim, full.styles + (size_t)(i + 1) * style_dim);
    }
    // one host thread per device: enqueue the shard (the call returns once its kernels are queued; the host waits only for the shard's
    // own integer durations), then record the event the gather waits on
    std::vector<std::thread> th;
    for (int r = 0; r < ndev; ++r)
        th.emplace_back([&, r] {
            Shard& s = sh[r];
            if (s.ids.empty()) return;
            try {
                sbv2_node::Dev& d = nd->devs[r];
                HIP_CHECK(hipSetDevice(d.device));
                VitsBatch v = full;
                v.n = (int)s.ids.size();
                v.t_lens = s.t_lens.data();
                v.phones = s.x.data();
                v.tones = s.tones.data();
                v.langs = s.langs.data();
                v.sids = s.sids.data();
                v.styles = s.styles.data();
                v.forced_durations = full.forced_durations ? s.forced.data() : nullptr;
                v.utt_ids = s.ids.data();
                pipeline_run_one(*d.bert->m, *d.vits->m, v, s.tok.data(), s.s_lens.data(), s.w2p.data());
                s.lens = d.vits->m->pcm_lens();
                HIP_CHECK(hipEventRecord(d.ev, d.vits->m->stream()));
                HIP_CHECK(hipStreamWaitEvent(d.xfer, d.ev, 0));
            } catch (const std::exception& e) {
                s.err = e.what();
            }
        });
    for (auto& t : th) t.join();
    for (auto& s : sh)
        if (!s.err.empty()) throw Error(s.err);

    // ---- gather to device 0: one message per device into `stage` (device order), then a permutation into utterance order -------------
    for (int r = 0; r < ndev; ++r)
        for (size_t j = 0; j < sh[r].ids.size(); ++j) pcm_lens[sh[r].ids[j]] = sh[r].lens[j];
    std::vector<int64_t> cnt, tab, doff(ndev + 1, 0);
    gather_plan(n, pcm_lens, rank_of.data(), ndev, cnt, tab);
    for (int r = 0; r < ndev; ++r) doff[r + 1] = doff[r] + cnt[r];
    const int64_t total = doff[ndev];
    SBV2_REQUIRE(total <= capacity, "PCM buffer too small: " + std::to_string(capacity) + " < " + std::to_string(total));
    sbv2_node::Dev& d0 = nd->devs[0];
    HIP_CHECK(hipSetDevice(d0.device));
    float* stage = static_cast<float*>(nd->stage.get(sizeof(float) * (size_t)std::max<int64_t>(total, 1)));
    float* ordered = static_cast<float*>(nd->ordered.get(sizeof(float) * (size_t)std::max<int64_t>(total, 1)));
    if (nd->use_rccl) {
        Rccl& R = rccl();
        NCCL_CHECK(R.GroupStart());
        for (int r = 1; r < ndev; ++r) {
            if (cnt[r] == 0) continue;
            HIP_CHECK(hipSetDevice(d0.device));
            NCCL_CHECK(R.Recv(stage + doff[r], (size_t)cnt[r], ncclFloat, r, d0.comm, d0.xfer));
            HIP_CHECK(hipSetDevice(nd->devs[r].device));
            NCCL_CHECK(R.Send(nd->devs[r].vits->m->pcm_device(), (size_t)cnt[r], ncclFloat, 0, nd->devs[r].comm, nd->devs[r].xfer));
        }
        NCCL_CHECK(R.GroupEnd());
        HIP_CHECK(hipSetDevice(d0.device));
    } else {
        for (int r = 1; r < ndev; ++r) {
            if (cnt[r] == 0) continue;
            sbv2_node::Dev& d = nd->devs[r];
            // the copy is queued on the SOURCE device's gather stream (it already waits for that shard's kernels); device 0 then waits on it
            HIP_CHECK(hipSetDevice(d.device));
            if (d.device == d0.device)
                HIP_CHECK(hipMemcpyAsync(stage + doff[r], d.vits->m->pcm_device(), sizeof(float) * (size_t)cnt[r], hipMemcpyDeviceToDevice, d.xfer));
            else
                HIP_CHECK(hipMemcpyPeerAsync(stage + doff[r], d0.device, d.vits->m->pcm_device(), d.device, sizeof(float) * (size_t)cnt[r], d.xfer));
            HIP_CHECK(hipEventRecord(d.ev, d.xfer));
            HIP_CHECK(hipSetDevice(d0.device));
            HIP_CHECK(hipStreamWaitEvent(d0.xfer, d.ev, 0));
        }
        HIP_CHECK(hipSetDevice(d0.device));
    }
    if (cnt[0] > 0)
        HIP_CHECK(hipMemcpyAsync(stage, d0.vits->m->pcm_device(), sizeof(float) * (size_t)cnt[0], hipMemcpyDeviceToDevice, d0.xfer));
    // permutation into the caller's utterance order (gather_plan's table)
    int64_t* d_tab = static_cast<int64_t*>(nd->table.get(sizeof(int64_t) * tab.size()));
    HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), sizeof(int64_t) * tab.size(), hipMemcpyHostToDevice, d0.xfer));
    copy_segments(stage, ordered, d_tab, n, d0.xfer);
    HIP_CHECK(hipGetLastError());
    if (total > 0) HIP_CHECK(hipMemcpyAsync(pcm_host, ordered, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost, d0.xfer));
    HIP_CHECK(hipStreamSynchronize(d0.xfer));
    // senders' buffers must stay untouched until their transfers have completed
    for (int r = 1; r < ndev; ++r) {
        HIP_CHECK(hipSetDevice(nd->devs[r].device));
        HIP_CHECK(hipStreamSynchronize(nd->devs[r].xfer));
    }
    HIP_CHECK(hipSetDevice(d0.device));
    API_END
}

}  // extern "C"
