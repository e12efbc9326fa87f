/* oracle/bow_oracle.c -- CPU oracle for the DBoW2 vocabulary path.  TEST INFRASTRUCTURE.
 *
 * Plain-C restatement of thirdparty/DBoW2/DBoW2 (paths relative to /root/reference):
 *   TemplatedVocabulary::loadFromTextFile   TemplatedVocabulary.h:1337-1420
 *   TemplatedVocabulary::transform (single) TemplatedVocabulary.h:1217-1259
 *   TemplatedVocabulary::transform (all)    TemplatedVocabulary.h:1126-1194
 *   BowVector::addWeight / normalize        BowVector.cpp:34-84
 *   FeatureVector::addFeature               FeatureVector.cpp:31-45
 *   L1Scoring::score                        ScoringObject.cpp:23-60
 *   FORB::distance                          FORB.cpp:81-101
 * Pinned against the REAL reference code for the BowVector / FeatureVector part:
 * oracle/Makefile.ref compiles those two reference .cpp files (they need no OpenCV) into
 * oracle/_ref/libdbow2_ref.so and tests/test_bow.py compares bit-for-bit.  The tree descent
 * lives in TemplatedVocabulary.h/FORB.h, which include OpenCV headers -> not buildable here.
 * Loader deviation (same as the product): empty lines are skipped.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int parent, nchildren, ccap, word_id;
    int* children;
    double weight;
    uint8_t desc[32];
} ov_node;

typedef struct orc_vocab {
    int k, L, scoring, weighting, nnodes, nwords, cap;
    ov_node* nodes;
} orc_vocab;

static int ov_add_node(orc_vocab* v)
{
    if (v->nnodes == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->nodes = (ov_node*)realloc(v->nodes, sizeof(ov_node) * v->cap); }
    memset(&v->nodes[v->nnodes], 0, sizeof(ov_node));
    return v->nnodes++;
}

void orc_vocab_free(orc_vocab* v)
{
    if (!v) return;
    for (int i = 0; i < v->nnodes; i++) free(v->nodes[i].children);
    free(v->nodes); free(v);
}

orc_vocab* orc_vocab_load_text(const char* path)
{
    FILE* f = fopen(path, "r");
    if (!f) return NULL;
    size_t cap = 1 << 16; char* line = (char*)malloc(cap);
    orc_vocab* v = (orc_vocab*)calloc(1, sizeof(*v));
    if (!fgets(line, (int)cap, f) || sscanf(line, "%d %d %d %d", &v->k, &v->L, &v->scoring, &v->weighting) != 4 ||
        v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || v->scoring < 0 || v->scoring > 5 ||
        v->weighting < 0 || v->weighting > 3) {                                   /* :1361-1366 */
        fclose(f); free(line); orc_vocab_free(v); return NULL;
    }
    ov_add_node(v);                                                               /* root, id 0 (:1376-1377) */
    while (fgets(line, (int)cap, f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
        if (!*p) continue;                                                        /* deviation: skip empty lines */
        const int nid = ov_add_node(v);
        char* e;
        const int pid = (int)strtol(p, &e, 10); p = e;
        const int isLeaf = (int)strtol(p, &e, 10); p = e;
        v->nodes[nid].parent = pid;
        ov_node* par = &v->nodes[pid];                                            /* children.push_back(nid) */
        if (par->nchildren == par->ccap) { par->ccap = par->ccap ? 2 * par->ccap : 16; par->children = (int*)realloc(par->children, sizeof(int) * par->ccap); }
        par->children[par->nchildren++] = nid;
        for (int i = 0; i < 32; i++) { v->nodes[nid].desc[i] = (uint8_t)strtol(p, &e, 10); p = e; }   /* FORB::fromString */
        v->nodes[nid].weight = strtod(p, &e);
        if (isLeaf > 0) v->nodes[nid].word_id = v->nwords++;                      /* :1408-1413 */
    }
    fclose(f); free(line);
    return v;
}

void orc_vocab_info(const orc_vocab* v, int* k, int* L, int* nnodes, int* nwords)
{ *k = v->k; *L = v->L; *nnodes = v->nnodes; *nwords = v->nwords; }

static int forb_distance(const uint8_t* a, const uint8_t* b)      /* FORB.cpp:81-101 */
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb; memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t x = pa ^ pb;
        x = x - ((x >> 1) & 0x55555555);
        x = (x & 0x33333333) + ((x >> 2) & 0x33333333);
        dist += (((x + (x >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* TemplatedVocabulary.h:1217-1259 */
void orc_bow_transform_one(const orc_vocab* v, const uint8_t* feature, int levelsup,
                           uint32_t* word_id, double* weight, uint32_t* nid)
{
    const int nid_level = v->L - levelsup;
    *nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const ov_node* nd = &v->nodes[final_id];
        final_id = nd->children[0];
        double best_d = forb_distance(feature, v->nodes[final_id].desc);
        for (int j = 1; j < nd->nchildren; j++) {
            const int id = nd->children[j];
            const double d = forb_distance(feature, v->nodes[id].desc);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) *nid = (uint32_t)final_id;
    } while (v->nodes[final_id].nchildren != 0);
    *word_id = (uint32_t)v->nodes[final_id].word_id;
    *weight = v->nodes[final_id].weight;
}

/* sorted-array std::map stand-ins */
static int lower_bound_u32(const uint32_t* a, int n, uint32_t key)
{ int lo = 0, hi = n; while (lo < hi) { int m = (lo + hi) / 2; if (a[m] < key) lo = m + 1; else hi = m; } return lo; }

/* TemplatedVocabulary.h:1126-1194 for weighting TF_IDF/TF/IDF/BINARY and any scoring type.
 * Outputs: bow_id/bow_val (ascending ids), fv as CSR (fv_node ascending, fv_start[nfv+1], fv_feat). */
int orc_bow_transform(const orc_vocab* v, const uint8_t* desc, int n, int levelsup,
                      uint32_t* bow_id, double* bow_val, int* n_bow,
                      uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_feat, int* n_fv)
{
    int nb = 0, nf = 0;
    uint32_t* f_node = (uint32_t*)malloc(sizeof(uint32_t) * (n + 1));    /* per accepted feature */
    uint32_t* f_idx = (uint32_t*)malloc(sizeof(uint32_t) * (n + 1));
    int nacc = 0;
    const int tf = (v->weighting == 0 || v->weighting == 1);
    for (int i = 0; i < n; i++) {
        uint32_t id, nid; double w;
        orc_bow_transform_one(v, desc + 32 * (size_t)i, levelsup, &id, &w, &nid);
        if (w > 0) {
            int p = lower_bound_u32(bow_id, nb, id);
            if (p < nb && bow_id[p] == id) { if (tf) bow_val[p] += w; }           /* addWeight / addIfNotExist */
            else {
                memmove(bow_id + p + 1, bow_id + p, sizeof(uint32_t) * (nb - p));
                memmove(bow_val + p + 1, bow_val + p, sizeof(double) * (nb - p));
                bow_id[p] = id; bow_val[p] = w; nb++;
            }
            f_node[nacc] = nid; f_idx[nacc] = (uint32_t)i; nacc++;
        }
    }
    /* mustNormalize, ScoringObject.h:74-89: every scoring type but DOT_PRODUCT(5); the norm is L2 for
     * L2_NORM(1) and L1 for L1_NORM(0), CHI_SQUARE(2), KL(3), BHATTACHARYYA(4) */
    const int must = (v->scoring != 5), l2 = (v->scoring == 1);
    if (tf && nb > 0 && !must) { const double nd = nb; for (int i = 0; i < nb; i++) bow_val[i] /= nd; }
    if (must) {                                                                    /* BowVector::normalize */
        double norm = 0.0;
        if (!l2) for (int i = 0; i < nb; i++) norm += fabs(bow_val[i]);
        else { for (int i = 0; i < nb; i++) norm += bow_val[i] * bow_val[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < nb; i++) bow_val[i] /= norm;
    }
    /* FeatureVector: group by node ascending, features in order of appearance */
    for (int i = 0; i < nacc; i++) {
        int p = lower_bound_u32(fv_node, nf, f_node[i]);
        if (!(p < nf && fv_node[p] == f_node[i])) {
            memmove(fv_node + p + 1, fv_node + p, sizeof(uint32_t) * (nf - p));
            fv_node[p] = f_node[i]; nf++;
        }
    }
    int pos = 0;
    for (int g = 0; g < nf; g++) {
        fv_start[g] = pos;
        for (int i = 0; i < nacc; i++) if (f_node[i] == fv_node[g]) fv_feat[pos++] = f_idx[i];
    }
    fv_start[nf] = pos;
    *n_bow = nb; *n_fv = nf;
    free(f_node); free(f_idx);
    return 0;
}

/* ScoringObject.cpp:23-60 */
double orc_bow_score_l1(const uint32_t* id1, const double* v1, int n1, const uint32_t* id2, const double* v2, int n2)
{
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (id1[i] == id2[j]) { score += fabs(v1[i] - v2[j]) - fabs(v1[i]) - fabs(v2[j]); i++; j++; }
        else if (id1[i] < id2[j]) i = lower_bound_u32(id1, n1, id2[j]);
        else j = lower_bound_u32(id2, n2, id1[i]);
    }
    return -score / 2.0;
}
