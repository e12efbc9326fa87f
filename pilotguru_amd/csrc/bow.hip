// bow.hip -- K8: DBoW2 vocabulary: flat blob, text loader, greedy tree descent on the GPU.
//
// Restates thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
//   loadFromTextFile :1337-1420 (header "k L scoring weighting", then one node per line:
//     parent isLeaf d0..d31 weight; node ids in file order, children appended in file order,
//     word ids in file order of the lines flagged as leaves)
//   transform(feature, word_id, weight, nid, levelsup) :1217-1259 (greedy descent, strict '<'
//     so the first minimum child wins; the node at level L - levelsup is reported)
//   transform(features, BowVector, FeatureVector, levelsup) :1126-1194 and BowVector.cpp:34-84,
//     FeatureVector.cpp:31-45, ScoringObject.cpp:23-60 for the host-side accumulation.
// FORB::distance (FORB.cpp:81-101) is the 256-bit Hamming distance.
//
// Blob layout (little endian, sections 64-byte aligned; identical to pilotguru_amd/vocab.py):
//   int32 header[16] = {'PGVC', version, k, L, nnodes, nwords, scoring, weighting, 0...}
//   u8  desc[nnodes][32]; f64 weight[nnodes]; i32 parent[nnodes]; i32 child0[nnodes];
//   i32 nchild[nnodes]; i32 word[nnodes]; i32 children[nnodes-1]
// One lane per feature; the 32-byte query stays in 8 VGPRs, children are gathered through L2
// (node descriptors of one parent are 32-B records; 120 k distances per 2000-feature frame).
#include "pgorb_internal.h"
#include <sys/stat.h>
#include <unistd.h>
#include <string>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#define PGVC_MAGIC 0x43564750

struct pgorb_vocab {
    std::vector<uint8_t> blob;
    int k, L, nnodes, nwords, scoring, weighting;
};

namespace {

size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

struct BlobView {
    const int32_t* hdr;
    const uint8_t* desc; const double* weight; const int32_t* parent; const int32_t* child0;
    const int32_t* nchild; const int32_t* word; const int32_t* children;
    size_t total;
};

bool view_blob(const uint8_t* b, size_t nbytes, BlobView& v)
{
    if (nbytes < 64) return false;
    v.hdr = reinterpret_cast<const int32_t*>(b);
    if (v.hdr[0] != PGVC_MAGIC || v.hdr[1] != 1) return false;
    const size_t n = (size_t)v.hdr[4];
    if (v.hdr[4] < 1) return false;
    size_t off = 64;
    v.desc = b + off; off = pad64(off + n * 32);
    v.weight = reinterpret_cast<const double*>(b + off); off = pad64(off + n * 8);
    v.parent = reinterpret_cast<const int32_t*>(b + off); off = pad64(off + n * 4);
    v.child0 = reinterpret_cast<const int32_t*>(b + off); off = pad64(off + n * 4);
    v.nchild = reinterpret_cast<const int32_t*>(b + off); off = pad64(off + n * 4);
    v.word = reinterpret_cast<const int32_t*>(b + off); off = pad64(off + n * 4);
    v.children = reinterpret_cast<const int32_t*>(b + off); off = pad64(off + (n - 1) * 4);
    v.total = off;
    if (nbytes < off) return false;
    // structure: a root with children, every child id in [1, n), child ranges inside children[]
    // (a truncated or corrupt blob -- they arrive from files and broadcasts -- is an error code, not a GPU fault)
    if (n < 2 || v.nchild[0] < 1) return false;
    for (size_t i = 0; i < n; i++) {
        const int64_t c0 = v.child0[i], nc = v.nchild[i];
        if (c0 < 0 || nc < 0 || c0 + nc > (int64_t)n - 1) return false;
        if (i && (v.parent[i] < 0 || (size_t)v.parent[i] >= n)) return false;
    }
    for (size_t i = 0; i + 1 < n; i++)
        if (v.children[i] < 1 || (size_t)v.children[i] >= n) return false;
    // child lists agree with the parent array: no cycle on any path from the root (see k_vocab_validate)
    for (size_t i = 0; i < n; i++)
        for (int64_t j = 0; j < v.nchild[i]; j++)
            if ((size_t)v.parent[v.children[v.child0[i] + j]] != i) return false;
    return true;
}

std::vector<uint8_t> pack_blob(int k, int L, int scoring, int weighting, const std::vector<uint8_t>& desc,
                               const std::vector<double>& weight, const std::vector<int32_t>& parent,
                               const std::vector<uint8_t>& leafFlag, int* nwords_out)
{
    const size_t n = parent.size();
    std::vector<int32_t> nchild(n, 0), child0(n, 0), word(n, -1), children(n > 1 ? n - 1 : 0);
    for (size_t i = 1; i < n; i++) nchild[parent[i]]++;
    for (size_t i = 1; i < n; i++) child0[i] = child0[i - 1] + nchild[i - 1];
    std::vector<int32_t> fill(n, 0);
    for (size_t i = 1; i < n; i++) { const int p = parent[i]; children[child0[p] + fill[p]++] = (int32_t)i; }
    int nwords = 0;
    for (size_t i = 1; i < n; i++) if (leafFlag[i]) word[i] = nwords++;     // :1408-1413
    *nwords_out = nwords;
    std::vector<uint8_t> out;
    auto put = [&](const void* p, size_t nb) {
        const size_t off = out.size();
        out.resize(pad64(off + nb), 0);
        if (nb) memcpy(out.data() + off, p, nb);
    };
    int32_t hdr[16] = {PGVC_MAGIC, 1, k, L, (int32_t)n, nwords, scoring, weighting};
    put(hdr, 64);
    put(desc.data(), n * 32); put(weight.data(), n * 8); put(parent.data(), n * 4);
    put(child0.data(), n * 4); put(nchild.data(), n * 4); put(word.data(), n * 4);
    put(children.data(), (n - 1) * 4);
    return out;
}

}  // namespace

// lives in api.hip
int pg_ctx_vocab_store(pgorb_ctx* c, const void* src, size_t nbytes, bool src_on_device, hipStream_t s);
int pg_ctx_vocab_get(pgorb_ctx* c, const uint8_t** d_blob, int* k, int* L, int* nnodes);
void pg_ctx_vocab_drop(pgorb_ctx* c);
int pg_ctx_fail(pgorb_ctx* c, int code, const char* msg);
int pg_ctx_stage(pgorb_ctx* c, int which, size_t bytes, void** p);
int pg_ctx_device(pgorb_ctx* c);

// structural checks of a blob that is already on the device (the host path runs view_blob): child ranges inside
// children[], child ids and parents inside [0, n), and every child list agrees with the parent array
// (parent[children[c0 + j]] == i).  With child ids >= 1 that rules out cycles on any path from the root: a node is
// only ever entered from its one parent, and the root is never entered -- so k_bow_transform's descent terminates
// on every blob that passes (round-3 advisory: in-range but cyclic links used to pass and hang the kernel).
// *bad != 0 when any node violates them
__global__ __launch_bounds__(256) void k_vocab_validate(const uint8_t* __restrict__ blob, int n, int* __restrict__ bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto pad = [](size_t v) { return (v + 63) / 64 * 64; };
    size_t off = 64;
    off = pad(off + (size_t)n * 32); off = pad(off + (size_t)n * 8);
    const int32_t* parent = reinterpret_cast<const int32_t*>(blob + off); off = pad(off + (size_t)n * 4);
    const int32_t* child0 = reinterpret_cast<const int32_t*>(blob + off); off = pad(off + (size_t)n * 4);
    const int32_t* nchild = reinterpret_cast<const int32_t*>(blob + off); off = pad(off + (size_t)n * 4);
    off = pad(off + (size_t)n * 4);
    const int32_t* children = reinterpret_cast<const int32_t*>(blob + off);
    const long long c0 = child0[i], nc = nchild[i];
    bool ok = c0 >= 0 && nc >= 0 && c0 + nc <= (long long)n - 1;
    if (i && (parent[i] < 0 || parent[i] >= n)) ok = false;
    if (i == 0 && nc < 1) ok = false;
    if (i + 1 < n && (children[i] < 1 || children[i] >= n)) ok = false;
    if (ok)
        for (long long j = 0; j < nc; j++) {
            const int ch = children[c0 + j];
            if (ch < 1 || ch >= n || parent[ch] != i) { ok = false; break; }
        }
    if (!ok) atomicExch(bad, 1);
}


__global__ __launch_bounds__(64) void k_bow_transform(const uint8_t* __restrict__ blob, int nnodes, int L,
                                                       const uint8_t* __restrict__ desc, int n, int levelsup,
                                                       uint32_t* __restrict__ word, double* __restrict__ weight,
                                                       uint32_t* __restrict__ node)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const size_t N = (size_t)nnodes;
    size_t off = 64;
    const uint8_t* vdesc = blob + off; off = (off + N * 32 + 63) / 64 * 64;
    const double* vweight = reinterpret_cast<const double*>(blob + off); off = (off + N * 8 + 63) / 64 * 64;
    off = (off + N * 4 + 63) / 64 * 64;                                        // parent
    const int32_t* child0 = reinterpret_cast<const int32_t*>(blob + off); off = (off + N * 4 + 63) / 64 * 64;
    const int32_t* nchild = reinterpret_cast<const int32_t*>(blob + off); off = (off + N * 4 + 63) / 64 * 64;
    const int32_t* vword = reinterpret_cast<const int32_t*>(blob + off); off = (off + N * 4 + 63) / 64 * 64;
    const int32_t* children = reinterpret_cast<const int32_t*>(blob + off);

    const uint4 q0 = reinterpret_cast<const uint4*>(desc + (int64_t)i * 32)[0];
    const uint4 q1 = reinterpret_cast<const uint4*>(desc + (int64_t)i * 32)[1];
    const int nid_level = L - levelsup;                     // :1225
    uint32_t nid = 0;                                       // root when nid_level <= 0
    int cur = 0, level = 0;
    do {                                                    // :1231-1254
        ++level;
        const int c0 = child0[cur], nc = nchild[cur];
        if (nc <= 0) break;                                 // (only a malformed blob: the loaders refuse a childless root)
        int best = -1, bestd = 0x7fffffff;
        for (int j = 0; j < nc; j++) {
            const int id = children[c0 + j];
            const uint4 d0 = reinterpret_cast<const uint4*>(vdesc + (int64_t)id * 32)[0];
            const uint4 d1 = reinterpret_cast<const uint4*>(vdesc + (int64_t)id * 32)[1];
            const int d = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                          __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
            if (d < bestd) { bestd = d; best = id; }        // strict '<': first minimum wins (:1244)
        }
        cur = best;
        if (level == nid_level) nid = (uint32_t)cur;
    } while (nchild[cur] > 0 && level < nnodes);            // isLeaf() == children.empty() (:328); (the bound: belt and braces, the validators rule cycles out)
    word[i] = (uint32_t)vword[cur];
    weight[i] = vweight[cur];
    node[i] = nid;
}

extern "C" {

int pgorb_vocab_load_text(const char* path, pgorb_vocab** out)
{
    if (!path || !out) return PGORB_E_ARG;
    *out = nullptr;
    // ORBvoc.txt is ~145 MB of decimal text (1.1 M lines of 35 numbers): the file is read in one piece and
    // tokenised in place (strtol / strtod on a NUL-terminated buffer); iostream extraction took 3.2 s for it.
    FILE* fp = fopen(path, "rb");
    if (!fp) return PGORB_E_ARG;
    std::vector<char> buf;
    {
        if (fseek(fp, 0, SEEK_END) != 0) { fclose(fp); return PGORB_E_ARG; }
        const long sz = ftell(fp);
        if (sz < 0 || fseek(fp, 0, SEEK_SET) != 0) { fclose(fp); return PGORB_E_ARG; }
        buf.resize((size_t)sz + 1);
        const size_t got = fread(buf.data(), 1, (size_t)sz, fp);
        fclose(fp);
        buf[got] = 0;
        buf.resize(got + 1);
    }
    char* p = buf.data();
    char* const end = buf.data() + buf.size() - 1;
    auto line_end = [&](char* q) { while (q < end && *q != '\n') q++; return q; };
    int k = -1, L = -1, n1 = -1, n2 = -1;
    {
        char* le = line_end(p);
        const char saved = *le; *le = 0;
        if (sscanf(p, "%d %d %d %d", &k, &L, &n1, &n2) != 4) return PGORB_E_ARG;
        *le = saved;
        p = le < end ? le + 1 : end;
    }
    if (k < 2 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return PGORB_E_ARG;   // :1361-1366
    std::vector<uint8_t> desc(32, 0), leaf(1, 0);
    std::vector<double> weight(1, 0.0);
    std::vector<int32_t> parent(1, -1);
    {
        size_t expect = 1, lv = 1;
        for (int l = 0; l < L; l++) { lv *= (size_t)k; expect += lv; }       // :1371-1373 reserves the same bound
        expect = std::min(expect, (size_t)(end - p) / 70 + 16);
        desc.reserve(expect * 32); leaf.reserve(expect); weight.reserve(expect); parent.reserve(expect);
    }
    while (p < end) {
        char* le = line_end(p);
        char* q = p;
        while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
        if (q == le) { p = le < end ? le + 1 : end; continue; }                // empty line: documented deviation
        const char saved = *le; *le = 0;                                       // numbers never run past the line
        char* e = nullptr;
        const long pid = strtol(q, &e, 10); q = e;
        const long isLeaf = strtol(q, &e, 10); q = e;
        const int nid = (int)parent.size();
        if (pid < 0 || pid >= nid) return PGORB_E_ARG;
        parent.push_back((int32_t)pid);
        leaf.push_back(isLeaf > 0);
        desc.resize(desc.size() + 32);
        uint8_t* dd = desc.data() + (size_t)nid * 32;
        for (int i = 0; i < 32; i++) { const long v = strtol(q, &e, 10); q = e; dd[i] = (uint8_t)v; }   // FORB::fromString
        const double w = strtod(q, &e);                                       // (a missing number reads as 0, like operator>>)
        weight.push_back(w);
        *le = saved;
        p = le < end ? le + 1 : end;
    }
    if (parent.size() < 2) return PGORB_E_ARG;             // header only: no vocabulary (the reference's transform would return an empty vector)
    pgorb_vocab* v = new pgorb_vocab();
    v->k = k; v->L = L; v->scoring = n1; v->weighting = n2; v->nnodes = (int)parent.size();
    v->blob = pack_blob(k, L, n1, n2, desc, weight, parent, leaf, &v->nwords);
    *out = v;
    return 0;
}

int pgorb_vocab_from_blob(const void* blob, int64_t nbytes, pgorb_vocab** out)
{
    if (!blob || !out) return PGORB_E_ARG;
    BlobView bv;
    if (!view_blob((const uint8_t*)blob, (size_t)nbytes, bv)) return PGORB_E_ARG;
    pgorb_vocab* v = new pgorb_vocab();
    v->blob.assign((const uint8_t*)blob, (const uint8_t*)blob + bv.total);
    v->k = bv.hdr[2]; v->L = bv.hdr[3]; v->nnodes = bv.hdr[4]; v->nwords = bv.hdr[5];
    v->scoring = bv.hdr[6]; v->weighting = bv.hdr[7];
    *out = v;
    return 0;
}

// ORBVocabulary(text file) with a binary cache beside the text: `<path>.pgvoc` = 32-byte header (magic -- its last character is
// the format version --, the text file's size and modification time in ns, the blob's length) + the flat blob.  A cache that matches the text file is
// loaded instead of parsing 145 MB of decimals (1.2 s for ORBvoc.txt; the blob is 67 MB); anything else -- no cache, stale,
// truncated, malformed -- parses the text and rewrites the cache (temp file + rename; failures to write are ignored).
// *from_cache (may be NULL) tells which way it went.
int pgorb_vocab_load_cached(const char* path, pgorb_vocab** out, int* from_cache)
{
    if (!path || !out) return PGORB_E_ARG;
    *out = nullptr;
    if (from_cache) *from_cache = 0;
    struct stat st;
    if (stat(path, &st) != 0) return PGORB_E_ARG;
    const int64_t srcSize = (int64_t)st.st_size, srcMtime = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    const std::string cpath = std::string(path) + ".pgvoc";
    const int64_t MAGIC = 0x32434F5647504750ll;                  // "PGPGVOC2": the trailing digit is the cache format's version word
    if (FILE* fp = fopen(cpath.c_str(), "rb")) {
        int64_t hdr[4] = {0, 0, 0, 0};
        struct stat cst;
        // the blob length is bounded by what the cache file actually holds (a corrupt or hostile header must not size an allocation)
        if (fstat(fileno(fp), &cst) == 0 && fread(hdr, 8, 4, fp) == 4 && hdr[0] == MAGIC && hdr[1] == srcSize && hdr[2] == srcMtime &&
            hdr[3] > 64 && hdr[3] <= (int64_t)cst.st_size - 32) {
            try {
                std::vector<uint8_t> blob((size_t)hdr[3]);
                if (fread(blob.data(), 1, blob.size(), fp) == blob.size() && pgorb_vocab_from_blob(blob.data(), (int64_t)blob.size(), out) == 0) {
                    fclose(fp);
                    if (from_cache) *from_cache = 1;
                    return 0;
                }
            } catch (const std::bad_alloc&) { *out = nullptr; }   // no exception crosses the C ABI: fall back to the text parse
        }
        fclose(fp);
    }
    const int rc = pgorb_vocab_load_text(path, out);
    if (rc) return rc;
    const std::string tmp = cpath + ".tmp" + std::to_string((long long)getpid());
    if (FILE* fp = fopen(tmp.c_str(), "wb")) {
        const int64_t hdr[4] = {MAGIC, srcSize, srcMtime, (int64_t)(*out)->blob.size()};
        const bool ok = fwrite(hdr, 8, 4, fp) == 4 && fwrite((*out)->blob.data(), 1, (*out)->blob.size(), fp) == (*out)->blob.size();
        if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), cpath.c_str()) != 0) (void)remove(tmp.c_str());
    }
    return 0;
}

int pgorb_vocab_blob(const pgorb_vocab* v, const void** blob, int64_t* nbytes)
{
    if (!v || !blob || !nbytes) return PGORB_E_ARG;
    *blob = v->blob.data(); *nbytes = (int64_t)v->blob.size();
    return 0;
}

int pgorb_vocab_info(const pgorb_vocab* v, int* k, int* L, int* nnodes, int* nwords, int* scoring, int* weighting)
{
    if (!v) return PGORB_E_ARG;
    if (k) *k = v->k; if (L) *L = v->L; if (nnodes) *nnodes = v->nnodes; if (nwords) *nwords = v->nwords;
    if (scoring) *scoring = v->scoring; if (weighting) *weighting = v->weighting;
    return 0;
}

void pgorb_vocab_free(pgorb_vocab* v) { delete v; }

int pgorb_vocab_upload(pgorb_ctx* c, const pgorb_vocab* v)
{
    if (!c || !v) return PGORB_E_ARG;
    return pg_ctx_vocab_store(c, v->blob.data(), v->blob.size(), false, 0);
}

// A blob that arrives on the device (a broadcast) never passed view_blob's structural checks on this rank: the same
// checks as a kernel, so that a corrupt blob is an error code here and not a fault inside k_bow_transform
int pg_vocab_validate_resident(pgorb_ctx* c, hipStream_t stream)
{
    const uint8_t* blob; int k, L, nn;
    int rc = pg_ctx_vocab_get(c, &blob, &k, &L, &nn);
    if (rc) return rc;
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    void* flag;
    if ((rc = pg_ctx_stage(c, 2, 64, &flag))) return rc;
    if (hipMemsetAsync(flag, 0, 4, stream) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipMemsetAsync failed");
    hipLaunchKernelGGL(k_vocab_validate, dim3((nn + 255) / 256), dim3(256), 0, stream, blob, nn, (int*)flag);
    int bad = 0;
    if (hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "vocabulary validation failed to run");
    if (bad) {
        pg_ctx_vocab_drop(c);
        return pg_ctx_fail(c, PGORB_E_ARG, "vocabulary blob is structurally invalid (child ranges / ids out of bounds)");
    }
    return 0;
}

int pgorb_vocab_upload_device(pgorb_ctx* c, const void* d_blob, int64_t nbytes, void* stream)
{
    if (!c || !d_blob || nbytes < 64) return PGORB_E_ARG;
    int rc = pg_ctx_vocab_store(c, d_blob, (size_t)nbytes, true, (hipStream_t)stream);
    if (rc) return rc;
    return pg_vocab_validate_resident(c, (hipStream_t)stream);
}

int pgorb_bow_transform_device(pgorb_ctx* c, const uint8_t* d_desc, int n, int levelsup, uint32_t* d_word,
                               double* d_weight, uint32_t* d_node, void* stream)
{
    if (!c) return PGORB_E_ARG;
    const uint8_t* blob; int k, L, nn;
    int rc = pg_ctx_vocab_get(c, &blob, &k, &L, &nn);
    if (rc) return rc;
    if (n < 0 || (n && (!d_desc || !d_word || !d_weight || !d_node)))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_bow_transform_device");
    if (!n) return 0;
    if (hipSetDevice(pg_ctx_device(c)) != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "hipSetDevice failed");
    hipLaunchKernelGGL(k_bow_transform, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, blob, nn, L,
                       d_desc, n, levelsup, d_word, d_weight, d_node);
    if (hipGetLastError() != hipSuccess) return pg_ctx_fail(c, PGORB_E_HIP, "k_bow_transform launch failed");
    return 0;
}

int pgorb_bow_transform(pgorb_ctx* c, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight,
                        uint32_t* node)
{
    if (!c) return PGORB_E_ARG;
    if (n < 0 || (n && (!desc || !word || !weight || !node)))
        return pg_ctx_fail(c, PGORB_E_ARG, "bad argument to pgorb_bow_transform");
    if (!n) return 0;
    void *dd, *dw, *dwt, *dn;
    int rc;
    if ((rc = pg_ctx_stage(c, 0, (size_t)n * 32, &dd))) return rc;
    if ((rc = pg_ctx_stage(c, 1, (size_t)n * 16 + 64, &dw))) return rc;
    if ((rc = pg_ctx_stage(c, 2, (size_t)n * 8 + 64, &dwt))) return rc;
    dn = (uint8_t*)dw + (size_t)n * 4 + (8 - ((size_t)n * 4) % 8) % 8;
    if (hipMemcpy(dd, desc, (size_t)n * 32, hipMemcpyHostToDevice) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy H2D failed");
    if ((rc = pgorb_bow_transform_device(c, (uint8_t*)dd, n, levelsup, (uint32_t*)dw, (double*)dwt, (uint32_t*)dn, 0))) return rc;
    if (hipMemcpy(word, dw, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(weight, dwt, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(node, dn, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return pg_ctx_fail(c, PGORB_E_HIP, "hipMemcpy D2H failed");
    return 0;
}

int pgorb_bow_vectors(int n, const uint32_t* word, const double* weight, const uint32_t* node, int scoring,
                      int weighting, uint32_t* bow_id, double* bow_val, int* n_bow, uint32_t* fv_node,
                      int32_t* fv_start, uint32_t* fv_feat, int* n_fv)
{
    if (n < 0 || !n_bow || !n_fv || (n && (!word || !weight || !node || !bow_id || !bow_val || !fv_node ||
                                           !fv_start || !fv_feat)))
        return PGORB_E_ARG;
    std::map<uint32_t, double> v;                          // BowVector
    std::map<uint32_t, std::vector<uint32_t>> fv;          // FeatureVector
    const bool tf = (weighting == 0 || weighting == 1);    // TF_IDF or TF (:1147)
    for (int i = 0; i < n; i++) {
        if (weight[i] > 0) {                               // not stopped (:1157 / :1185)
            if (tf) v[word[i]] += weight[i];               // BowVector::addWeight (same double adds, in feature order)
            else v.insert(std::make_pair(word[i], weight[i]));   // addIfNotExist
            fv[node[i]].push_back((uint32_t)i);            // FeatureVector::addFeature
        }
    }
    // mustNormalize (ScoringObject.h:74-89): L1_NORM(0), CHI_SQUARE(2), KL(3), BHATTACHARYYA(4) -> true with the
    // L1 norm; L2_NORM(1) -> true with the L2 norm; DOT_PRODUCT(5) -> false
    const bool must = (scoring != 5);
    const bool l2 = (scoring == 1);
    if (tf && !v.empty() && !must) {                       // :1164-1170
        const double nd = (double)v.size();
        for (auto& kv : v) kv.second /= nd;
    }
    if (must) {                                            // BowVector::normalize (:62-84)
        double norm = 0.0;
        if (!l2) for (auto& kv : v) norm += fabs(kv.second);
        else { for (auto& kv : v) norm += kv.second * kv.second; norm = sqrt(norm); }
        if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
    }
    int nb = 0;
    for (auto& kv : v) { bow_id[nb] = kv.first; bow_val[nb] = kv.second; nb++; }
    *n_bow = nb;
    int nf = 0, pos = 0;
    for (auto& kv : fv) {
        fv_node[nf] = kv.first; fv_start[nf] = pos;
        for (uint32_t f : kv.second) fv_feat[pos++] = f;
        nf++;
    }
    if (n) fv_start[nf] = pos;
    *n_fv = nf;
    return 0;
}

double pgorb_bow_score_l1(const uint32_t* id1, const double* val1, int n1, const uint32_t* id2, const double* val2, int n2)
{
    // L1Scoring::score, ScoringObject.cpp:23-60: only common words contribute
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (id1[i] == id2[j]) { score += fabs(val1[i] - val2[j]) - fabs(val1[i]) - fabs(val2[j]); i++; j++; }
        else if (id1[i] < id2[j]) i++;
        else j++;
    }
    return -score / 2.0;
}

}  // extern "C"
