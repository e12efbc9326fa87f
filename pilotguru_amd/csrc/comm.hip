// comm.hip -- the ONE collective of the path: the ORB vocabulary, parsed once, broadcast to every GPU of the node.
//
// The reference is a single process that loads the vocabulary once and shares it by pointer
// (src/optical_trajectories.cc:87-94: `ORB_SLAM2::ORBVocabulary vocabulary(FLAGS_vocabulary_file)` handed to every new System;
// thirdparty/orb-slam2/src/ORBVocabulary.cc:7-9).  With one extractor context per GPU the pointer share becomes ONE
// ncclBroadcast of the flat blob (bow.hip) over xGMI -- RCCL called directly, behind the C ABI (SURVEY.md section 7 step 6,
// section 8(b)/(e)).  Nothing else of the path communicates: frames and rides are sharded, results are gathered on the host.
//
// Two ways to form the group, one broadcast call:
//   pgorb_comm_create_local   ONE process, one host thread per device (the CLI's --devices=0,1,...): ncclCommInitAll over
//                             the DISTINCT devices of the contexts; contexts that share a device share its rank and get
//                             the blob by a device-to-device copy from the rank's buffer (so eight contexts on one GPU
//                             -- the one-box test of BASELINE config 4 -- go through the same entry point);
//   pgorb_comm_create_rank    one PROCESS per GPU (bench.py under torch.distributed.run): ncclCommInitRank with a
//                             unique id from pgorb_comm_unique_id that the launcher's control plane hands round.
// The receive buffer IS the context's vocabulary arena: the collective writes the blob where k_bow_transform reads it.
// A received blob never passed the host loader's structural checks on that rank, so every receiver runs
// k_vocab_validate before the vocabulary counts as resident (a corrupt blob is an error code, not a fault).
//
// librccl is opened on first use (dlopen): a process that never broadcasts does not load it, and libpgorb.so keeps
// HIP + libstdc++ as its only link-time dependencies.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 67 MB
// ORBvoc blob is one message, bound by a single link's rate, ~0.5 ms on the wire -- start-up, not the per-frame path.
#include "pgorb_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <string>
#include <vector>

int pg_ctx_vocab_reserve(pgorb_ctx* c, size_t nbytes, void** p);           // api.hip
int pg_ctx_vocab_commit(pgorb_ctx* c, size_t nbytes, hipStream_t s);
int pg_ctx_vocab_get(pgorb_ctx* c, const uint8_t** d_blob, int* k, int* L, int* nnodes);
int pg_ctx_vocab_store(pgorb_ctx* c, const void* src, size_t nbytes, bool src_on_device, hipStream_t s);
int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg);
int pg_ctx_device(pgorb_ctx* c);
extern "C" int pg_vocab_validate_resident(pgorb_ctx* c, hipStream_t stream);   // bow.hip

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    std::string path;         // the file ncclGetUniqueId resolved into (dladdr)
    bool preloaded = false;   // the process had it mapped before the library asked
};

// The RCCL the library calls must be the one the PROCESS already has: under torch.distributed the process holds torch's
// bundled torch/lib/librccl.so, and a second build loaded by name would hand the 128-byte ncclUniqueId of one RCCL to the
// other.  So: first ask the loader for a librccl that is ALREADY mapped (RTLD_NOLOAD, by soname and by the path
// /proc/self/maps shows), and only when the process has none load one by name.
void* rccl_already_mapped(std::string* path)
{
    const char* sonames[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : sonames)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) { *path = n; return h; }
    // a copy loaded by path whose soname the loader does not index under these names (torch/lib/librccl.so)
    if (FILE* f = fopen("/proc/self/maps", "r")) {
        char line[4096];
        void* h = nullptr;
        while (!h && fgets(line, sizeof line, f)) {
            const char* p = strchr(line, '/');
            if (!p) continue;
            std::string file(p);
            while (!file.empty() && (file.back() == '\n' || file.back() == ' ')) file.pop_back();
            const size_t base = file.rfind('/');
            if (file.compare(base + 1, 10, "librccl.so") != 0) continue;
            if ((h = dlopen(file.c_str(), RTLD_NOW | RTLD_NOLOAD))) *path = file;
        }
        fclose(f);
        return h;
    }
    return nullptr;
}

Rccl* rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string lastErr;
        if (const char* forced = getenv("PGORB_RCCL_LIBRARY"); forced && *forced) {
            if (!(R.so = dlopen(forced, RTLD_NOW | RTLD_LOCAL))) { const char* e = dlerror(); lastErr = e ? e : "dlopen failed"; }
        }
        if (!R.so && (R.so = rccl_already_mapped(&R.path))) R.preloaded = true;
        if (!R.so) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) {
                if ((R.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
                const char* e = dlerror();                             // (dlerror() clears the message: read it ONCE per failure)
                lastErr = e ? e : "dlopen failed";
            }
        }
        if (!R.so) { R.error = "librccl.so not found (" + lastErr + ")"; return; }
#define PG_SYM(f) if (!(R.f = reinterpret_cast<decltype(R.f)>(dlsym(R.so, "nccl" #f)))) R.error = "librccl lacks nccl" #f
        PG_SYM(GetUniqueId); PG_SYM(CommInitRank); PG_SYM(CommInitAll); PG_SYM(CommDestroy); PG_SYM(Broadcast);
        PG_SYM(GroupStart); PG_SYM(GroupEnd); PG_SYM(GetErrorString);
#undef PG_SYM
        Dl_info info;
        if (R.GetUniqueId && dladdr(reinterpret_cast<void*>(R.GetUniqueId), &info) && info.dli_fname) R.path = info.dli_fname;
    });
    return &R;
}

}  // namespace

struct pgorb_comm {
    // local form: ranks = distinct devices in order of first appearance; rank form: exactly one entry (this process's)
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<int> devices;
    std::vector<pgorb_ctx*> ctxs;        // every member context
    std::vector<int> rankOfCtx;          // index into comms/devices (local form); 0 in rank form
    std::vector<int> leaderOfRank;       // the context whose arena is the rank's send / receive buffer
    bool local = true;
    int myRank = 0, nranks = 1;          // rank form
    void* d_count = nullptr;             // rank form: 8 bytes on the device for the byte count
};

namespace {

// the calling thread's current device is put back when a group call returns (the calls walk the group's devices)
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

int comm_fail(pgorb_comm* cm, int code, const std::string& msg)
{
    return pg_ctx_fail(cm->ctxs.empty() ? nullptr : cm->ctxs[0], code, msg.c_str());
}

int nccl_fail(pgorb_comm* cm, const char* what, ncclResult_t r)
{
    return comm_fail(cm, PGORB_E_HIP, std::string(what) + ": " + rccl()->GetErrorString(r));
}

}  // namespace

extern "C" {

// Which librccl the library resolved (the file ncclGetUniqueId lives in), for diagnostics: bench.py prints it, the tests
// assert it is the one the process already held.  Returns 0, or PGORB_E_HIP when no librccl could be opened (the text is
// then the loader's message).  *preloaded (may be NULL): 1 when the process had it mapped before libpgorb asked.
int pgorb_comm_library(char* path, int cap, int* preloaded)
{
    Rccl* R = rccl();
    const std::string& t = R->error.empty() ? R->path : R->error;
    if (path && cap > 0) { strncpy(path, t.c_str(), (size_t)cap - 1); path[cap - 1] = 0; }
    if (preloaded) *preloaded = R->preloaded ? 1 : 0;
    return R->error.empty() ? 0 : PGORB_E_HIP;
}

int pgorb_comm_unique_id(void* id)
{
    if (!id) return PGORB_E_ARG;
    Rccl* R = rccl();
    if (!R->error.empty()) return PGORB_E_HIP;
    static_assert(sizeof(ncclUniqueId) == PGORB_COMM_ID_BYTES, "pgorb.h's id size is RCCL's");
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return PGORB_E_HIP;
    memcpy(id, &u, sizeof u);
    return 0;
}

int pgorb_comm_create_local(pgorb_ctx* const* ctxs, int nctx, pgorb_comm** out)
{
    if (!ctxs || nctx < 1 || !out) return PGORB_E_ARG;
    *out = nullptr;
    for (int i = 0; i < nctx; i++) if (!ctxs[i]) return PGORB_E_ARG;
    Rccl* R = rccl();
    if (!R->error.empty()) return pg_ctx_fail(ctxs[0], PGORB_E_HIP, R->error.c_str());
    DeviceRestore keep;
    pgorb_comm* cm = new pgorb_comm();
    cm->local = true;
    for (int i = 0; i < nctx; i++) {
        const int dev = pg_ctx_device(ctxs[i]);
        int r = -1;
        for (size_t k = 0; k < cm->devices.size(); k++) if (cm->devices[k] == dev) r = (int)k;
        if (r < 0) { r = (int)cm->devices.size(); cm->devices.push_back(dev); cm->leaderOfRank.push_back(i); }
        cm->ctxs.push_back(ctxs[i]);
        cm->rankOfCtx.push_back(r);
    }
    cm->nranks = (int)cm->devices.size();
    cm->comms.assign(cm->nranks, nullptr);
    const ncclResult_t r = R->CommInitAll(cm->comms.data(), cm->nranks, cm->devices.data());
    if (r != ncclSuccess) { const int rc = nccl_fail(cm, "ncclCommInitAll", r); cm->comms.clear(); pgorb_comm_destroy(cm); return rc; }
    for (int k = 0; k < cm->nranks; k++) {
        hipStream_t s = nullptr;
        if (hipSetDevice(cm->devices[k]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            const int rc = comm_fail(cm, PGORB_E_HIP, "stream for the broadcast");
            pgorb_comm_destroy(cm);
            return rc;
        }
        cm->streams.push_back(s);
    }
    *out = cm;
    return 0;
}

int pgorb_comm_create_rank(pgorb_ctx* ctx, int rank, int nranks, const void* id, pgorb_comm** out)
{
    if (!ctx || !out || !id || nranks < 1 || rank < 0 || rank >= nranks) return PGORB_E_ARG;
    *out = nullptr;
    Rccl* R = rccl();
    if (!R->error.empty()) return pg_ctx_fail(ctx, PGORB_E_HIP, R->error.c_str());
    DeviceRestore keep;
    pgorb_comm* cm = new pgorb_comm();
    cm->local = false; cm->myRank = rank; cm->nranks = nranks;
    cm->ctxs.push_back(ctx); cm->rankOfCtx.push_back(0); cm->leaderOfRank.push_back(0);
    cm->devices.push_back(pg_ctx_device(ctx));
    cm->comms.assign(1, nullptr);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    hipStream_t s = nullptr;
    bool ok = hipSetDevice(cm->devices[0]) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
    if (ok) cm->streams.push_back(s);                                  // (owned by the group from here on: destroy releases it)
    if (!ok || hipMalloc(&cm->d_count, 64) != hipSuccess) {
        const int rc = comm_fail(cm, PGORB_E_HIP, "stream for the broadcast");
        cm->comms.clear();
        pgorb_comm_destroy(cm);
        return rc;
    }
    const ncclResult_t r = R->CommInitRank(&cm->comms[0], nranks, u, rank);
    if (r != ncclSuccess) { const int rc = nccl_fail(cm, "ncclCommInitRank", r); cm->comms.clear(); pgorb_comm_destroy(cm); return rc; }
    *out = cm;
    return 0;
}

void pgorb_comm_destroy(pgorb_comm* cm)
{
    if (!cm) return;
    DeviceRestore keep;
    Rccl* R = rccl();
    for (size_t k = 0; k < cm->streams.size(); k++) {
        (void)hipSetDevice(cm->devices[k]);
        (void)hipStreamSynchronize(cm->streams[k]);
        (void)hipStreamDestroy(cm->streams[k]);
    }
    for (ncclComm_t c : cm->comms) if (c && R->CommDestroy) (void)R->CommDestroy(c);
    if (cm->d_count) { (void)hipSetDevice(cm->devices[0]); (void)hipFree(cm->d_count); }
    delete cm;
}

int pgorb_comm_ranks(const pgorb_comm* cm) { return cm ? cm->nranks : PGORB_E_ARG; }

// ORBVocabulary shared with every System (src/optical_trajectories.cc:87-94) -> one ncclBroadcast.
//   local form: `root` indexes the contexts the group was created from, `v` is the parsed vocabulary (host);
//   rank form:  `root` is a rank; `v` is read on the root rank only (the others pass NULL).
// On return the vocabulary is resident (and validated) in every member context of this process.  *seconds (may be
// NULL): the wall time of the collective itself, upload to the root and validation excluded.
int pgorb_vocab_broadcast(pgorb_comm* cm, int root, const pgorb_vocab* v, double* seconds)
{
    if (!cm) return PGORB_E_ARG;
    if (seconds) *seconds = 0.0;
    DeviceRestore keep;
    Rccl* R = rccl();
    const int nctx = (int)cm->ctxs.size();
    if (cm->local ? (root < 0 || root >= nctx) : (root < 0 || root >= cm->nranks)) return comm_fail(cm, PGORB_E_ARG, "broadcast root out of range");
    const bool iAmRoot = cm->local || cm->myRank == root;
    if (iAmRoot && !v) return comm_fail(cm, PGORB_E_ARG, "the root of the vocabulary broadcast needs the vocabulary");
    int rc;
    if (cm->local) {
        // the root context's arena is its rank's buffer: make it the leader of its device
        const int rootRank = cm->rankOfCtx[root];
        cm->leaderOfRank[rootRank] = root;
        if ((rc = pgorb_vocab_upload(cm->ctxs[root], v))) return rc;
        const void* blob; int64_t nbytes;
        if ((rc = pgorb_vocab_blob(v, &blob, &nbytes))) return rc;
        std::vector<void*> buf(cm->nranks, nullptr);
        for (int k = 0; k < cm->nranks; k++) {
            pgorb_ctx* lead = cm->ctxs[cm->leaderOfRank[k]];
            if (k == rootRank) { const uint8_t* b; int kk, L, nn; if ((rc = pg_ctx_vocab_get(lead, &b, &kk, &L, &nn))) return rc; buf[k] = const_cast<uint8_t*>(b); }
            else if ((rc = pg_ctx_vocab_reserve(lead, (size_t)nbytes, &buf[k]))) return rc;
        }
        for (int k = 0; k < cm->nranks; k++) { (void)hipSetDevice(cm->devices[k]); (void)hipStreamSynchronize(cm->streams[k]); }
        const auto t0 = std::chrono::steady_clock::now();
        ncclResult_t r = R->GroupStart();
        if (r != ncclSuccess) return nccl_fail(cm, "ncclGroupStart", r);
        bool devFail = false;
        for (int k = 0; k < cm->nranks && r == ncclSuccess; k++) {
            if (hipSetDevice(cm->devices[k]) != hipSuccess) { devFail = true; break; }      // (the group is ended below, whatever happened)
            r = R->Broadcast(buf[rootRank], buf[k], (size_t)nbytes, ncclUint8, rootRank, cm->comms[k], cm->streams[k]);
        }
        const ncclResult_t re = R->GroupEnd();
        if (devFail) return comm_fail(cm, PGORB_E_HIP, "hipSetDevice failed");
        if (r != ncclSuccess) return nccl_fail(cm, "ncclBroadcast", r);
        if (re != ncclSuccess) return nccl_fail(cm, "ncclGroupEnd", re);
        for (int k = 0; k < cm->nranks; k++) {
            if (hipSetDevice(cm->devices[k]) != hipSuccess || hipStreamSynchronize(cm->streams[k]) != hipSuccess)
                return comm_fail(cm, PGORB_E_HIP, "the vocabulary broadcast did not complete");
        }
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // receivers: header + structure checks; then the other contexts of each device copy from their rank's buffer
        for (int k = 0; k < cm->nranks; k++) {
            if (k == rootRank) continue;
            pgorb_ctx* lead = cm->ctxs[cm->leaderOfRank[k]];
            if ((rc = pg_ctx_vocab_commit(lead, (size_t)nbytes, cm->streams[k]))) return rc;
            if ((rc = pg_vocab_validate_resident(lead, cm->streams[k]))) return rc;
        }
        for (int i = 0; i < nctx; i++) {
            const int k = cm->rankOfCtx[i];
            if (i == cm->leaderOfRank[k]) continue;
            if ((rc = pgorb_vocab_upload_device(cm->ctxs[i], buf[k], nbytes, cm->streams[k]))) return rc;
        }
        return 0;
    }
    // ---- one process per GPU ----
    pgorb_ctx* c = cm->ctxs[0];
    hipStream_t s = cm->streams[0];
    if (hipSetDevice(cm->devices[0]) != hipSuccess) return comm_fail(cm, PGORB_E_HIP, "hipSetDevice failed");
    int64_t nbytes = 0;
    const void* blob = nullptr;
    if (iAmRoot) {
        if ((rc = pgorb_vocab_upload(c, v))) return rc;
        if ((rc = pgorb_vocab_blob(v, &blob, &nbytes))) return rc;
        if (hipMemcpyAsync(cm->d_count, &nbytes, 8, hipMemcpyHostToDevice, s) != hipSuccess) return comm_fail(cm, PGORB_E_HIP, "byte count upload");
    }
    // message 1: the byte count (receivers size their arena), message 2: the blob
    ncclResult_t r = R->Broadcast(cm->d_count, cm->d_count, 8, ncclUint8, root, cm->comms[0], s);
    if (r != ncclSuccess) return nccl_fail(cm, "ncclBroadcast (byte count)", r);
    if (hipMemcpyAsync(&nbytes, cm->d_count, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return comm_fail(cm, PGORB_E_HIP, "byte count of the vocabulary broadcast");
    if (nbytes < 64 || nbytes > ((int64_t)1 << 40)) return comm_fail(cm, PGORB_E_ARG, "vocabulary broadcast announced an impossible size");
    void* buf = nullptr;
    if (iAmRoot) { const uint8_t* b; int kk, L, nn; if ((rc = pg_ctx_vocab_get(c, &b, &kk, &L, &nn))) return rc; buf = const_cast<uint8_t*>(b); }
    else if ((rc = pg_ctx_vocab_reserve(c, (size_t)nbytes, &buf))) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    r = R->Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, cm->comms[0], s);
    if (r != ncclSuccess) return nccl_fail(cm, "ncclBroadcast", r);
    if (hipStreamSynchronize(s) != hipSuccess) return comm_fail(cm, PGORB_E_HIP, "the vocabulary broadcast did not complete");
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!iAmRoot) {
        if ((rc = pg_ctx_vocab_commit(c, (size_t)nbytes, s))) return rc;
        if ((rc = pg_vocab_validate_resident(c, s))) return rc;
    }
    return 0;
}

}  // extern "C"
