// oracle/ref_dbow2_wrap.cpp -- C entry points over the REFERENCE's own DBoW2 translation units
// (BowVector.cpp and FeatureVector.cpp compiled where they lie under /root/reference; they
// include no OpenCV.  ScoringObject.cpp includes TemplatedVocabulary.h -> OpenCV -> not buildable).  TEST INFRASTRUCTURE: pins the oracle's and the
// product's BowVector / FeatureVector arithmetic to the real reference code.
#include <stdint.h>
#include <vector>
#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// Replays TemplatedVocabulary::transform's accumulation (TemplatedVocabulary.h:1147-1193) for
// TF_IDF weighting + L1 scoring on per-feature (word, weight, node) triples, calling the
// reference BowVector::addWeight / normalize and FeatureVector::addFeature.
int ref_bow_vectors(int n, const uint32_t* word, const double* weight, const uint32_t* node,
                    uint32_t* bow_id, double* bow_val, int* n_bow,
                    uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_feat, int* n_fv)
{
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) { v.addWeight(word[i], weight[i]); fv.addFeature(node[i], (unsigned)i); }
    v.normalize(DBoW2::L1);
    int nb = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) { bow_id[nb] = it->first; bow_val[nb] = it->second; nb++; }
    int nf = 0, pos = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        fv_node[nf] = it->first; fv_start[nf] = pos;
        for (size_t k = 0; k < it->second.size(); k++) fv_feat[pos++] = it->second[k];
        nf++;
    }
    fv_start[nf] = pos;
    *n_bow = nb; *n_fv = nf;
    return 0;
}

// The same replay for any (scoring, weighting) pair of the vocabulary header: TF_IDF(0) / TF(1) accumulate
// with addWeight, IDF(2) / BINARY(3) with addIfNotExist (TemplatedVocabulary.h:1147-1190); the norm comes from
// the reference's own table of scoring classes, ScoringObject.h:74-89 (the classes themselves need
// ScoringObject.cpp -> TemplatedVocabulary.h -> OpenCV, so the table is restated here, the arithmetic is
// the reference's BowVector::normalize).
int ref_bow_vectors2(int n, const uint32_t* word, const double* weight, const uint32_t* node, int scoring, int weighting,
                     uint32_t* bow_id, double* bow_val, int* n_bow,
                     uint32_t* fv_node, int32_t* fv_start, uint32_t* fv_feat, int* n_fv)
{
    static const bool kMust[6] = {true, true, true, true, true, false};
    static const DBoW2::LNorm kNorm[6] = {DBoW2::L1, DBoW2::L2, DBoW2::L1, DBoW2::L1, DBoW2::L1, DBoW2::L1};
    if (scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return -1;
    const bool must = kMust[scoring];
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    if (weighting == DBoW2::TF || weighting == DBoW2::TF_IDF) {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) { v.addWeight(word[i], weight[i]); fv.addFeature(node[i], (unsigned)i); }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) { v.addIfNotExist(word[i], weight[i]); fv.addFeature(node[i], (unsigned)i); }
    }
    if (must) v.normalize(kNorm[scoring]);
    int nb = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it) { bow_id[nb] = it->first; bow_val[nb] = it->second; nb++; }
    int nf = 0, pos = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        fv_node[nf] = it->first; fv_start[nf] = pos;
        for (size_t k = 0; k < it->second.size(); k++) fv_feat[pos++] = it->second[k];
        nf++;
    }
    fv_start[nf] = pos;
    *n_bow = nb; *n_fv = nf;
    return 0;
}

}  // extern "C"
