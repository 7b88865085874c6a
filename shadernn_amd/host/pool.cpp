// pool.cpp -- multi-device runner below the language bindings (SURVEY 8b "Threading": one context + host thread per device; 8e: batch split,
// weights replicated, no data-path collective, RCCL only at the edges).
//
// The reference is single-device (core/src/ic2/vulkanBackend.cpp:30-31 creates one device, MixedInferenceCore::run is not re-entrant); this is
// the MI355X-side design for the 8 GPUs of a node.  A pool holds one REPLICA per entry of `devices` (an entry may repeat: two contexts on one GPU).
// A replica is a host thread that owns everything of its device: its HipContext + stream (created in that thread, so the thread's current HIP
// device is the replica's for every call it ever makes), and one MixedInferenceCore per micro-batch slot of its share
//     images [g*B/G, (g+1)*B/G)  of a global batch of B images over G replicas              (shadernn_amd/dist.py shard_range's rule).
// snn_pool_run releases all replicas together; each enqueues `steps` passes over its slots (RunParameters::deferSync) and waits once; the call
// returns when the slowest is done -- the in-process equivalent of bench.py's barrier / MAX-over-ranks under torch.distributed.  Outputs stay
// sharded on their devices; snn_pool_download_output gathers them through host memory (each replica copies its own images into the caller's
// buffer), snn_pool_allgather_output_rccl does it with one ncclAllGather per slot over xGMI (librccl.so is dlopen'ed: the library has no link-time
// dependency on it) and hands rank 0's copy to the host -- for classifier logits; never on the timed path.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/snn_c.h"
#include "../../include/snnhip.h"
#include "snn/contextFactory.h"
#include "snn/core.h"
#include "snn/utils.h"

namespace {

struct Replica {
    int index = 0, device = 0;
    int first = 0, images = 0;          // the shard
    std::vector<int> slotFirst, slotImages; // micro-batch slots of the shard (global image index, count)
    std::vector<snn_model*> models;     // one per slot
    std::thread thread;
    // task hand-off
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;
    bool hasTask = false, done = true, quit = false;
    int rc = 0;
};

} // namespace

struct snn_pool {
    std::vector<Replica*> reps;
    int globalBatch = 0, inW = 0, inH = 0, inC = 0;
    int outHWC[3] = {0, 0, 0};
    bool half = false;
    // RCCL (lazily initialised by snn_pool_allgather_output_rccl)
    void* rccl = nullptr;
    std::vector<void*> comms;
    std::vector<snnhip_ctx*> ctxs;           // borrowed from the replicas' first model (for the gather buffers)
    std::vector<snnhip_tensor*> gatherBufs; // [replica]: G * slotImages images of the output
};

namespace {

void workerLoop(Replica* r) {
    for (;;) {
        std::function<int()> t;
        {
            std::unique_lock<std::mutex> lk(r->m);
            r->cv.wait(lk, [&] { return r->hasTask || r->quit; });
            if (r->quit && !r->hasTask) return;
            t = std::move(r->task);
            r->hasTask = false;
        }
        int rc = -2;
        try {
            rc = t();
        } catch (...) {
        }
        {
            std::lock_guard<std::mutex> lk(r->m);
            r->rc = rc;
            r->done = true;
        }
        r->cv.notify_all();
    }
}

void post(Replica* r, std::function<int()> f) {
    std::lock_guard<std::mutex> lk(r->m);
    r->task = std::move(f);
    r->hasTask = true;
    r->done = false;
    r->cv.notify_all();
}

int wait(Replica* r) {
    std::unique_lock<std::mutex> lk(r->m);
    r->cv.wait(lk, [&] { return r->done; });
    return r->rc;
}

// the same task on every replica, in parallel; first non-zero return code (or 0)
int onAll(snn_pool* p, const std::function<int(Replica*)>& f) {
    for (Replica* r : p->reps) post(r, [f, r] { return f(r); });
    int rc = 0;
    for (Replica* r : p->reps) {
        const int k = wait(r);
        if (k != 0 && rc == 0) rc = k;
    }
    return rc;
}

size_t imageFloatsIn(const snn_pool* p) { return static_cast<size_t>(p->inH) * p->inW * p->inC; }
size_t imageFloatsOut(const snn_pool* p) { return static_cast<size_t>(p->outHWC[0]) * p->outHWC[1] * p->outHWC[2]; }

} // namespace

extern "C" {

int snn_pool_create(const char* json_path, const int* devices, int n_devices, int in_w, int in_h, int in_c, int prefer_half, int capture_graph,
                    int global_batch, int micro_batch, snn_pool** out) {
    if (!json_path || !devices || !out || n_devices < 1 || global_batch < 1 || in_w < 1 || in_h < 1 || in_c < 1) return -1;
    // every device must exist BEFORE a replica thread is started: a context failure inside MixedInferenceCore is fatal (SNN_RIP aborts the process,
    // the reference's convention), a bad device list is an error code
    // (probed on a thread of its own: the current HIP device is per thread, and the caller's must be what it was when this call returns)
    int probeRc = 0;
    std::thread([&] {
        for (int g = 0; g < n_devices && probeRc == 0; ++g) {
            snnhip_ctx* probe = nullptr;
            if (snnhip_ctx_create(devices[g], &probe) != SNNHIP_OK) {
                SNN_LOGE("snn_pool_create: device %d: %s", devices[g], snnhip_last_error());
                probeRc = -2;
            } else {
                snnhip_ctx_destroy(probe);
            }
        }
    }).join();
    if (probeRc != 0) return probeRc;
    auto* p = new snn_pool();
    p->globalBatch = global_batch;
    p->inW = in_w;
    p->inH = in_h;
    p->inC = in_c;
    p->half = prefer_half != 0;
    const std::string path = json_path;
    for (int g = 0; g < n_devices; ++g) {
        auto* r = new Replica();
        r->index = g;
        r->device = devices[g];
        // GPU g of G gets images [g*B/G, (g+1)*B/G)  (shadernn_amd/dist.py shard_range)
        r->first = static_cast<int>(static_cast<long long>(g) * global_batch / n_devices);
        r->images = static_cast<int>(static_cast<long long>(g + 1) * global_batch / n_devices) - r->first;
        const int mb = micro_batch > 0 ? micro_batch : (r->images > 0 ? r->images : 1);
        for (int at = 0; at < r->images; at += mb) {
            r->slotFirst.push_back(r->first + at);
            r->slotImages.push_back(r->images - at < mb ? r->images - at : mb);
        }
        r->thread = std::thread(workerLoop, r);
        p->reps.push_back(r);
    }
    const int rc = onAll(p, [=](Replica* r) {
        for (size_t s = 0; s < r->slotImages.size(); ++s) {
            snn_model* m = nullptr;
            const int k = snn_model_create4(path.c_str(), r->device, in_w, in_h, in_c, 0, 1, 0, prefer_half, capture_graph, r->slotImages[s], &m);
            if (k != 0) return k;
            r->models.push_back(m);
        }
        return 0;
    });
    if (rc != 0) {
        snn_pool_destroy(p);
        return rc;
    }
    for (Replica* r : p->reps)
        if (!r->models.empty()) {
            snn_model_output_dims(r->models[0], p->outHWC);
            break;
        }
    *out = p;
    return 0;
}

int snn_pool_destroy(snn_pool* p) {
    if (!p) return 0;
    // everything a replica built is destroyed by the thread that built it (its HIP device is that thread's current device)
    onAll(p, [p](Replica* r) {
        if (static_cast<size_t>(r->index) < p->gatherBufs.size() && p->gatherBufs[r->index]) snnhip_tensor_free(p->gatherBufs[r->index]);
        for (snn_model* m : r->models) snn_model_destroy(m);
        r->models.clear();
        return 0;
    });
    if (p->rccl) {
        using DestroyFn = int (*)(void*);
        auto destroy = reinterpret_cast<DestroyFn>(dlsym(p->rccl, "ncclCommDestroy"));
        for (void* c : p->comms)
            if (c && destroy) destroy(c);
        dlclose(p->rccl);
    }
    for (Replica* r : p->reps) {
        {
            std::lock_guard<std::mutex> lk(r->m);
            r->quit = true;
        }
        r->cv.notify_all();
        if (r->thread.joinable()) r->thread.join();
        delete r;
    }
    delete p;
    return 0;
}

int snn_pool_replicas(snn_pool* p) { return p ? static_cast<int>(p->reps.size()) : 0; }

int snn_pool_shard(snn_pool* p, int replica, int* first_image, int* images, int* slots) {
    if (!p || replica < 0 || replica >= static_cast<int>(p->reps.size())) return -1;
    if (first_image) *first_image = p->reps[replica]->first;
    if (images) *images = p->reps[replica]->images;
    if (slots) *slots = static_cast<int>(p->reps[replica]->slotImages.size());
    return 0;
}

int snn_pool_output_dims(snn_pool* p, int hwc[3]) {
    if (!p || !hwc) return -1;
    memcpy(hwc, p->outHWC, sizeof(p->outHWC));
    return 0;
}

int snn_pool_upload_input(snn_pool* p, const float* nhwc) {
    if (!p || !nhwc) return -1;
    const size_t per = imageFloatsIn(p);
    return onAll(p, [=](Replica* r) {
        for (size_t s = 0; s < r->models.size(); ++s)
            if (snn_model_upload_input(r->models[s], nhwc + per * static_cast<size_t>(r->slotFirst[s])) != 0) return -1;
        return 0;
    });
}

int snn_pool_run(snn_pool* p, int steps, double* seconds) {
    if (!p || steps < 1) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = onAll(p, [=](Replica* r) {
        for (int k = 0; k < steps; ++k)
            for (snn_model* m : r->models)
                if (snn_model_run_async(m) != 0) return -1;
        for (snn_model* m : r->models)
            if (snn_model_sync(m) != 0) return -1;
        return 0;
    });
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int snn_pool_download_output(snn_pool* p, float* nhwc) {
    if (!p || !nhwc) return -1;
    const size_t per = imageFloatsOut(p);
    return onAll(p, [=](Replica* r) {
        for (size_t s = 0; s < r->models.size(); ++s)
            if (snn_model_download_output(r->models[s], nhwc + per * static_cast<size_t>(r->slotFirst[s])) != 0) return -1;
        return 0;
    });
}

// ---- the edge collective: ncclAllGather of the (small) output tensors, e.g. classifier logits --------------------------------------
// rccl.h's types restated (the library is loaded at run time): ncclResult_t = int (0 = success), ncclComm_t = opaque pointer, ncclDataType_t
// ncclFloat16 = 6 / ncclFloat32 = 7 (rccl/rccl.h: ncclHalf, ncclFloat).
int snn_pool_allgather_output_rccl(snn_pool* p, float* nhwc_rank0) {
    if (!p || !nhwc_rank0) return -1;
    const int G = static_cast<int>(p->reps.size());
    for (int a = 0; a < G; ++a) {
        for (int b = a + 1; b < G; ++b)
            if (p->reps[a]->device == p->reps[b]->device) return -3; // RCCL refuses two ranks on one device: use snn_pool_download_output
        if (p->reps[a]->images != p->reps[0]->images || p->reps[a]->slotImages != p->reps[0]->slotImages) return -3; // allgather = equal shares
    }
    using InitAllFn = int (*)(void**, int, const int*);
    using GroupFn = int (*)();
    using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, void*);
    if (!p->rccl) {
        // the handle and the communicators are published only once ncclCommInitAll has succeeded: a failed attempt leaves the pool as it was and
        // the next call starts over
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
            SNN_LOGE("snn_pool_allgather_output_rccl: librccl.so not found (%s)", dlerror());
            return -4;
        }
        auto initAll = reinterpret_cast<InitAllFn>(dlsym(lib, "ncclCommInitAll"));
        std::vector<int> devs;
        for (Replica* r : p->reps) devs.push_back(r->device);
        std::vector<void*> comms(static_cast<size_t>(G), nullptr);
        if (!initAll || initAll(comms.data(), G, devs.data()) != 0) {
            SNN_LOGE("snn_pool_allgather_output_rccl: ncclCommInitAll over %d device(s) failed", G);
            dlclose(lib);
            return -4;
        }
        p->comms = comms;
        p->rccl = lib;
    }
    auto groupStart = reinterpret_cast<GroupFn>(dlsym(p->rccl, "ncclGroupStart"));
    auto groupEnd = reinterpret_cast<GroupFn>(dlsym(p->rccl, "ncclGroupEnd"));
    auto allGather = reinterpret_cast<AllGatherFn>(dlsym(p->rccl, "ncclAllGather"));
    if (!groupStart || !groupEnd || !allGather) return -4;
    const int slots = static_cast<int>(p->reps[0]->slotImages.size());
    const size_t per = imageFloatsOut(p);
    // receive buffers: one per replica, big enough for the largest slot of every rank
    int maxSlot = 0;
    for (int s = 0; s < slots; ++s) maxSlot = p->reps[0]->slotImages[s] > maxSlot ? p->reps[0]->slotImages[s] : maxSlot;
    if (p->gatherBufs.empty()) {
        p->gatherBufs.assign(static_cast<size_t>(G), nullptr);
        p->ctxs.assign(static_cast<size_t>(G), nullptr);
        const int rc = onAll(p, [=](Replica* r) {
            snnhip_ctx* ctx = snn_model_hip_ctx(r->models[0]);
            p->ctxs[r->index] = ctx;
            // (the output tensor's own type: a model asked for half precision may still end in an fp32 layer)
            return snnhip_tensor_alloc(ctx, G * maxSlot, p->outHWC[0], p->outHWC[1], p->outHWC[2], snnhip_tensor_dtype(snn_model_output_tensor(r->models[0])),
                                       &p->gatherBufs[r->index]);
        });
        if (rc != 0) return -2;
    }
    std::vector<float> host(static_cast<size_t>(G) * maxSlot * per);
    for (int s = 0; s < slots; ++s) {
        const int mb = p->reps[0]->slotImages[s];
        const size_t count = static_cast<size_t>(mb) * per; // elements per rank
        if (onAll(p, [=](Replica* r) { return snn_model_sync(r->models[s]); }) != 0) return -2;
        if (groupStart() != 0) return -4;
        bool sent = true;
        for (int g = 0; g < G && sent; ++g) {
            Replica* r = p->reps[g];
            const snnhip_tensor* o = snn_model_output_tensor(r->models[s]);
            const int ncclType = snnhip_tensor_dtype(o) == SNNHIP_F16 ? 6 : 7;
            sent = allGather(snnhip_tensor_data(o), snnhip_tensor_data(p->gatherBufs[g]), count, ncclType, p->comms[g], snnhip_ctx_stream(p->ctxs[g])) == 0;
        }
        const bool closed = groupEnd() == 0; // (always: an error between the two calls must not leave the group open)
        if (!sent || !closed) return -4;
        // rank 0's copy to the host (its stream orders the download behind the collective); rank g's block is images [slotFirst_g[s], +mb)
        if (onAll(p, [&](Replica* r) {
                if (r->index != 0) return snnhip_sync(p->ctxs[r->index]) == SNNHIP_OK ? 0 : -2;
                return snnhip_tensor_download(p->gatherBufs[0], host.data()) == SNNHIP_OK ? 0 : -2;
            }) != 0)
            return -2;
        for (int g = 0; g < G; ++g) memcpy(nhwc_rank0 + per * static_cast<size_t>(p->reps[g]->slotFirst[s]), host.data() + static_cast<size_t>(g) * count, count * sizeof(float));
    }
    return 0;
}

} // extern "C"
