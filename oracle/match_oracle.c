/* oracle/match_oracle.c -- CPU oracle for the Frame grid and ORBmatcher::SearchForInitialization.
 * TEST INFRASTRUCTURE (see orb_oracle.h).  Restates, line by line (paths under
 * /root/reference/thirdparty/orb-slam2):
 *   Frame::AssignFeaturesToGrid / PosInGrid     src/Frame.cc:234-249, 386-396
 *   Frame::GetFeaturesInArea                    src/Frame.cc:331-384
 *   ORBmatcher::SearchForInitialization         src/ORBmatcher.cc:407-522
 *   ORBmatcher::ComputeThreeMaxima              src/ORBmatcher.cc:1605-1646
 * for undistorted == raw keypoints (pilotguru's calibration path with k1 == 0,
 * src/Frame.cc:408-438) and image bounds (0, cols, 0, rows) (src/Frame.cc:461-466).
 */
#include "orb_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GRID_COLS 64          /* include/Frame.h:37-38 */
#define GRID_ROWS 48
#define HISTO_LENGTH 30       /* src/ORBmatcher.cc:38-40 */
#define TH_LOW 50

/* CSR grid: cell = ix*GRID_ROWS + iy; start[GRID_COLS*GRID_ROWS+1]; idx[n] keeps insertion order */
void orc_frame_grid(const orc_keypoint* kps, int n, float minX, float maxX, float minY, float maxY,
                    int32_t* start, int32_t* idx)
{
    const float invW = (float)GRID_COLS / (maxX - minX);         /* Frame.cc:216-217 */
    const float invH = (float)GRID_ROWS / (maxY - minY);
    const int ncell = GRID_COLS * GRID_ROWS;
    int* cell = (int*)malloc(sizeof(int) * (n + 1));
    memset(start, 0, sizeof(int32_t) * (ncell + 1));
    for (int i = 0; i < n; i++) {
        const int posX = (int)roundf((kps[i].x - minX) * invW);   /* PosInGrid, Frame.cc:388-389 */
        const int posY = (int)roundf((kps[i].y - minY) * invH);
        cell[i] = (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) ? -1 : posX * GRID_ROWS + posY;
        if (cell[i] >= 0) start[cell[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) start[c + 1] += start[c];
    int* fill = (int*)calloc(ncell, sizeof(int));
    for (int i = 0; i < n; i++)
        if (cell[i] >= 0) idx[start[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
}

/* Frame::GetFeaturesInArea, Frame.cc:331-384.  Returns the number of indices written. */
int orc_features_in_area(const orc_keypoint* kps, const int32_t* start, const int32_t* idx,
                         float minX, float maxX, float minY, float maxY,
                         float x, float y, float r, int minLevel, int maxLevel, int32_t* out)
{
    const float invW = (float)GRID_COLS / (maxX - minX);
    const float invH = (float)GRID_ROWS / (maxY - minY);
    int n = 0;
    int nMinCellX = (int)floorf((x - minX - r) * invW); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - minX + r) * invW); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - minY - r) * invH); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - minY + r) * invH); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = start[c]; j < start[c + 1]; j++) {
                const orc_keypoint* kp = &kps[idx[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out[n++] = idx[j];
            }
        }
    return n;
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1605-1646 (histogram given as bin sizes) */
static void three_maxima(const int* histo, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ORBmatcher::SearchForInitialization, ORBmatcher.cc:407-522.  prev_matched: float[2*n1] in/out
 * (vbPrevMatched); matches12: int32[n1] out.  Returns nmatches. */
int orc_search_for_initialization(const orc_keypoint* kps1, const uint8_t* desc1, int n1,
                                  const orc_keypoint* kps2, const uint8_t* desc2, int n2,
                                  const int32_t* grid2_start, const int32_t* grid2_idx,
                                  float minX, float maxX, float minY, float maxY,
                                  float* prev_matched, int32_t* matches12,
                                  int windowSize, float nnratio, int checkOrientation)
{
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int* rotBin = (int*)malloc(sizeof(int) * (n1 + 1));          /* bin an i1 was pushed to, or -1 */
    for (int i = 0; i < n1; i++) rotBin[i] = -1;
    const float factor = 1.0f / HISTO_LENGTH;
    int* vMatchedDistance = (int*)malloc(sizeof(int) * (n2 + 1));
    int* vnMatches21 = (int*)malloc(sizeof(int) * (n2 + 1));
    for (int i = 0; i < n2; i++) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (n2 + 1));

    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        const int nind = orc_features_in_area(kps2, grid2_start, grid2_idx, minX, maxX, minY, maxY,
                                              prev_matched[2 * i1], prev_matched[2 * i1 + 1],
                                              (float)windowSize, level1, level1, vIndices2);
        if (nind == 0) continue;
        const uint8_t* d1 = desc1 + 32 * (size_t)i1;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < nind; k++) {
            const int i2 = vIndices2[k];
            const int dist = orc_descriptor_distance(d1, desc2 + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) {
                    float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotBin[i1] = bin;
                }
            }
        }
    }
    if (checkOrientation) {
        int histo[HISTO_LENGTH] = {0};
        for (int i = 0; i < n1; i++) if (rotBin[i] >= 0) histo[rotBin[i]]++;
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i1 = 0; i1 < n1; i1++) {
            const int b = rotBin[i1];
            if (b < 0 || b == ind1 || b == ind2 || b == ind3) continue;
            if (matches12[i1] >= 0) { matches12[i1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                                /* :516-519 */
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = kps2[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y;
        }
    free(rotBin); free(vMatchedDistance); free(vnMatches21); free(vIndices2);
    return nmatches;
}

/* ---------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, float th)
 * src/ORBmatcher.cc:46-131, RadiusByViewingCos :133-139.  MapPoint fields as arrays:
 * valid = mbTrackInView && !isBad(); proj_x/y = mTrackProjX/Y; level = mnTrackScaleLevel;
 * view_cos = mTrackViewCos; pdesc = GetDescriptor(); pobs = Observations() > 0.
 * kp_has_point[i]: F.mvpMapPoints[i] holds a point with Observations() > 0 before the call.
 * assigned[i] = index of the map point written to F.mvpMapPoints[i] by this call, or -1. */
#define TH_HIGH 100
int orc_search_by_projection_points(const orc_keypoint* kps, const uint8_t* desc, int n,
                                    const int32_t* grid_start, const int32_t* grid_idx,
                                    float minX, float maxX, float minY, float maxY,
                                    const float* scale_factors, const uint8_t* kp_has_point,
                                    int npoints, const uint8_t* valid, const float* proj_x, const float* proj_y,
                                    const int32_t* level, const float* view_cos, const uint8_t* pdesc,
                                    const uint8_t* pobs, float th, float nnratio, int32_t* assigned)
{
    int nmatches = 0;
    uint8_t* taken = (uint8_t*)malloc(n + 1);
    for (int i = 0; i < n; i++) { taken[i] = kp_has_point ? (kp_has_point[i] != 0) : 0; assigned[i] = -1; }
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
    const int bFactor = th != 1.0;
    for (int iMP = 0; iMP < npoints; iMP++) {
        if (!valid[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = ((double)view_cos[iMP] > 0.998) ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        const int nind = orc_features_in_area(kps, grid_start, grid_idx, minX, maxX, minY, maxY, proj_x[iMP], proj_y[iMP],
                                              r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (nind == 0) continue;
        const uint8_t* d = pdesc + 32 * (size_t)iMP;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nind; k++) {
            const int idx = vIndices[k];
            if (taken[idx]) continue;
            const int dist = orc_descriptor_distance(d, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = kps[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assigned[bestIdx] = iMP;
            taken[bestIdx] = pobs[iMP] != 0;
            nmatches++;
        }
    }
    free(taken); free(vIndices);
    return nmatches;
}

/* The matching loop of ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame,
 * float th, bool bMono = true), src/ORBmatcher.cc:1355-1474, given the projections (u, v) of the
 * last frame's map points (valid = pMP && !outlier && invzc >= 0 && inside the bounds). */
int orc_search_by_projection_frame(const orc_keypoint* kps, const uint8_t* desc, int n,
                                   const int32_t* grid_start, const int32_t* grid_idx,
                                   float minX, float maxX, float minY, float maxY,
                                   const float* scale_factors, const uint8_t* kp_has_point,
                                   int nlast, const uint8_t* valid, const float* u, const float* v,
                                   const int32_t* last_octave, const float* last_angle, const uint8_t* pdesc,
                                   const uint8_t* pobs, float th, int checkOrientation, int32_t* assigned)
{
    int nmatches = 0;
    uint8_t* taken = (uint8_t*)malloc(n + 1);
    for (int i = 0; i < n; i++) { taken[i] = kp_has_point ? (kp_has_point[i] != 0) : 0; assigned[i] = -1; }
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
    int* histBin = (int*)malloc(sizeof(int) * (nlast + 1));       /* rotHist as (bin, bestIdx2) per push */
    int* histIdx = (int*)malloc(sizeof(int) * (nlast + 1));
    int npush = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nlast; i++) {
        if (!valid[i]) continue;
        const int nLastOctave = last_octave[i];
        const float radius = th * scale_factors[nLastOctave];
        const int nind = orc_features_in_area(kps, grid_start, grid_idx, minX, maxX, minY, maxY, u[i], v[i], radius,
                                              nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (nind == 0) continue;
        const uint8_t* dMP = pdesc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nind; k++) {
            const int i2 = vIndices2[k];
            if (taken[i2]) continue;
            const int dist = orc_descriptor_distance(dMP, desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            assigned[bestIdx2] = i;
            taken[bestIdx2] = pobs[i] != 0;
            nmatches++;
            if (checkOrientation) {
                float rot = last_angle[i] - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                histBin[npush] = bin; histIdx[npush] = bestIdx2; npush++;
            }
        }
    }
    if (checkOrientation) {
        int histo[HISTO_LENGTH] = {0};
        for (int k = 0; k < npush; k++) histo[histBin[k]]++;
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int k = 0; k < npush; k++)
            if (histBin[k] != ind1 && histBin[k] != ind2 && histBin[k] != ind3) { assigned[histIdx[k]] = -1; nmatches--; }
    }
    free(taken); free(vIndices2); free(histBin); free(histIdx);
    return nmatches;
}

/* MapPoint::PredictScale's logarithm (src/MapPoint.cc:524: `log(ratio)` on a float with TemplatedVocabulary.h:36's
 * `using namespace std` in effect = std::log(float) = the platform's logf; likewise Frame.cc:188 for mfLogScaleFactor).
 * PARITY CONTRACT (like orc_sincos_f): a fixed double-precision sequence, rounded once to float --
 *   x = m * 2^e, m in [sqrt(1/2), sqrt(2));  s = (m - 1) / (m + 1);  log x = e ln2 + 2 s (1 + s^2/3 + s^4/5 + ... + s^20/21)
 * (truncation < 2^-56 relative, no FMA) -- within 1 ulp of any libm's logf; tests/test_oracle.py measures the difference from
 * this box's glibc.  x <= 0 -> -inf (the reference would take the log of maxDistance / dist3D > 0), inf -> inf, NaN -> NaN. */
float orc_log_f(float xf)
{
    if (xf != xf) return xf;
    if (!(xf > 0.0f)) return -INFINITY;
    if (xf == INFINITY) return INFINITY;
    union { double d; uint64_t u; } v;
    v.d = (double)xf;                                             /* every positive float is a normal double */
    int e = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;   /* m in [1, 2) */
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }           /* m in (sqrt(1/2), sqrt(2)] */
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0; p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;  p = p * z + 1.0 / 7.0;  p = p * z + 1.0 / 5.0;  p = p * z + 1.0 / 3.0;  p = p * z + 1.0;
    const double r = (double)e * 0.6931471805599453 + (2.0 * s) * p;
    return (float)r;
}

/* MapPoint::PredictScale(currentDist, Frame*), src/MapPoint.cc:516-531.  `(int)ceil(...)` of a NaN or of a value outside
 * int's range is what x86-64's cvttss2si returns: INT_MIN, i.e. level 0 after the clamp. */
int orc_predict_scale(float max_distance, float current_dist, float log_scale_factor, int nlevels)
{
    const float ratio = max_distance / current_dist;
    const float q = ceilf(orc_log_f(ratio) / log_scale_factor);
    int nScale = (q != q || q >= 2147483648.0f || q < -2147483648.0f) ? INT_MIN : (int)q;
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}

/* The matching loop of ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*>
 * &sAlreadyFound, float th, int ORBdist), src/ORBmatcher.cc:1476-1603 (Tracking::Relocalization, Tracking.cc:1434,1448:
 * th 10 / ORBdist 100, then th 3 / ORBdist 64), given per key-frame map point i what the caller's pose arithmetic
 * (:1497-1517, cv::Mat) produced: valid = pMP && !pMP->isBad(); found = sAlreadyFound.count(pMP); (u, v) the projection;
 * dist3d = |x3Dw - Ow|; min / max_distance = GetMin/MaxDistanceInvariance(); kf_angle = pKF->mvKeysUn[i].angle.
 * kp_has_point[i2]: CurrentFrame.mvpMapPoints[i2] != NULL before the call (ANY point blocks, :1542-1543 -- no
 * Observations() test here).  assigned[i2] = i, or -1.  Returns nmatches. */
int orc_search_by_projection_keyframe(const orc_keypoint* kps, const uint8_t* desc, int n,
                                      const int32_t* grid_start, const int32_t* grid_idx,
                                      float minX, float maxX, float minY, float maxY,
                                      const float* scale_factors, int nlevels, float log_scale_factor, const uint8_t* kp_has_point,
                                      int npoints, const uint8_t* valid, const uint8_t* found, const float* u, const float* v,
                                      const float* dist3d, const float* min_distance, const float* max_distance,
                                      const float* kf_angle, const uint8_t* pdesc, float th, int ORBdist, int checkOrientation,
                                      int32_t* assigned)
{
    int nmatches = 0;
    uint8_t* taken = (uint8_t*)malloc(n + 1);
    for (int i = 0; i < n; i++) { taken[i] = kp_has_point ? (kp_has_point[i] != 0) : 0; assigned[i] = -1; }
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (n + 1));
    int* histBin = (int*)malloc(sizeof(int) * (npoints + 1));
    int* histIdx = (int*)malloc(sizeof(int) * (npoints + 1));
    int npush = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < npoints; i++) {
        if (!valid[i] || found[i]) continue;                                          /* :1497-1499 */
        if (u[i] < minX || u[i] > maxX) continue;                                     /* :1512-1515 */
        if (v[i] < minY || v[i] > maxY) continue;
        if (dist3d[i] < min_distance[i] || dist3d[i] > max_distance[i]) continue;     /* :1525-1526 */
        const int nPredictedLevel = orc_predict_scale(max_distance[i], dist3d[i], log_scale_factor, nlevels);   /* :1528 */
        const float radius = th * scale_factors[nPredictedLevel];                     /* :1531 */
        const int nind = orc_features_in_area(kps, grid_start, grid_idx, minX, maxX, minY, maxY, u[i], v[i], radius,
                                              nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
        if (nind == 0) continue;
        const uint8_t* dMP = pdesc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nind; k++) {
            const int i2 = vIndices2[k];
            if (taken[i2]) continue;                                                  /* :1542-1543 */
            const int dist = orc_descriptor_distance(dMP, desc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist && bestIdx2 >= 0) {      /* (ORBdist >= 256 with no candidate left would index -1 in the reference) */
            assigned[bestIdx2] = i;
            taken[bestIdx2] = 1;
            nmatches++;
            if (checkOrientation) {
                float rot = kf_angle[i] - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                histBin[npush] = bin; histIdx[npush] = bestIdx2; npush++;
            }
        }
    }
    if (checkOrientation) {
        int histo[HISTO_LENGTH] = {0};
        for (int k = 0; k < npush; k++) histo[histBin[k]]++;
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int k = 0; k < npush; k++)
            if (histBin[k] != ind1 && histBin[k] != ind2 && histBin[k] != ind3) { assigned[histIdx[k]] = -1; nmatches--; }
    }
    free(taken); free(vIndices2); free(histBin); free(histIdx);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches),
 * src/ORBmatcher.cc:161-290.  FeatureVectors as CSR (ascending nodes).  matches[j] = key-frame
 * keypoint whose map point went to vpMapPointMatches[j], or -1.  Returns nmatches. */
int orc_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_point_valid, int nkf,
                      const uint32_t* aNode, const int32_t* aStart, const uint32_t* aFeat, int nA,
                      const uint8_t* f_desc, const float* f_angle, int nf,
                      const uint32_t* bNode, const int32_t* bStart, const uint32_t* bFeat, int nB,
                      float nnratio, int checkOrientation, int32_t* matches)
{
    int nmatches = 0;
    for (int i = 0; i < nf; i++) matches[i] = -1;
    int* histBin = (int*)malloc(sizeof(int) * (nf + 1));
    int* histIdx = (int*)malloc(sizeof(int) * (nf + 1));
    int npush = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    int a = 0, b = 0;
    while (a < nA && b < nB) {
        if (aNode[a] == bNode[b]) {
            for (int ia = aStart[a]; ia < aStart[a + 1]; ia++) {
                const unsigned realIdxKF = aFeat[ia];
                if (!kf_point_valid[realIdxKF]) continue;
                const uint8_t* dKF = kf_desc + 32 * (size_t)realIdxKF;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int ib = bStart[b]; ib < bStart[b + 1]; ib++) {
                    const unsigned realIdxF = bFeat[ib];
                    if (matches[realIdxF] >= 0) continue;
                    const int dist = orc_descriptor_distance(dKF, f_desc + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        matches[bestIdxF] = (int32_t)realIdxKF;
                        if (checkOrientation) {
                            float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            histBin[npush] = bin; histIdx[npush] = bestIdxF; npush++;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (aNode[a] < bNode[b]) {
            while (a < nA && aNode[a] < bNode[b]) a++;                   /* lower_bound */
        } else {
            while (b < nB && bNode[b] < aNode[a]) b++;
        }
    }
    if (checkOrientation) {
        int histo[HISTO_LENGTH] = {0};
        for (int k = 0; k < npush; k++) histo[histBin[k]]++;
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int k = 0; k < npush; k++)
            if (histBin[k] != ind1 && histBin[k] != ind2 && histBin[k] != ind3) { matches[histIdx[k]] = -1; nmatches--; }
    }
    free(histBin); free(histIdx);
    return nmatches;
}

/* cv2.4: cv::undistortPoints(src, dst, K, distCoeffs, Mat(), K) (imgproc/undistort.cpp,
 * cvUndistortPoints: 5 fixed-point iterations in double), as called by Frame::UndistortKeyPoints
 * (src/Frame.cc:408-438) and Frame::ComputeImageBounds (:440-467).  PARITY UNPINNED (OpenCV). */
static void undistort_point(const double K[4], const double k[5], float px, float py, float* ox, float* oy)
{
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
    double x = px, y = py, x0, y0;
    x0 = x = (x - cx) * ifx;
    y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = fx * x + 0. * y + cx;
    double yy = 0. * x + fy * y + cy;
    double ww = 1. / (0. * x + 0. * y + 1.);
    *ox = (float)(xx * ww); *oy = (float)(yy * ww);
}
void orc_undistort_keypoints(const orc_keypoint* kps, int n, const float camera[4], const float dist[5], orc_keypoint* out)
{
    const double K[4] = {camera[0], camera[1], camera[2], camera[3]};
    const double k[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
    for (int i = 0; i < n; i++) {
        out[i] = kps[i];
        if (dist[0] != 0.0f) undistort_point(K, k, kps[i].x, kps[i].y, &out[i].x, &out[i].y);
    }
}
void orc_image_bounds(int cols, int rows, const float camera[4], const float dist[5], float bounds[4])
{
    if (dist[0] == 0.0f) { bounds[0] = 0; bounds[1] = (float)cols; bounds[2] = 0; bounds[3] = (float)rows; return; }
    const double K[4] = {camera[0], camera[1], camera[2], camera[3]};
    const double k[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
    float ux[4], uy[4];
    const float cxs[4] = {0, (float)cols, 0, (float)cols}, cys[4] = {0, 0, (float)rows, (float)rows};
    for (int i = 0; i < 4; i++) undistort_point(K, k, cxs[i], cys[i], &ux[i], &uy[i]);
    bounds[0] = fminf(ux[0], ux[2]); bounds[1] = fmaxf(ux[1], ux[3]);
    bounds[2] = fminf(uy[0], uy[1]); bounds[3] = fmaxf(uy[2], uy[3]);
}
