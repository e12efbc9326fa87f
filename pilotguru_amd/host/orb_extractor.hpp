// orb_extractor.hpp -- C++ host-side mirror of ORB_SLAM2::ORBextractor / ORBmatcher over the
// C ABI (include/pgorb.h).  Same names, argument meaning and error behaviour as
//   thirdparty/orb-slam2/include/ORBextractor.h:44-110   (operator(), Get* accessors)
//   thirdparty/orb-slam2/include/ORBmatcher.h:40-53      (DescriptorDistance, SearchForInitialization)
// but OpenCV-free: images are raw 8-bit planes, keypoints are pgorb_keypoint (the cv::KeyPoint
// layout), descriptors are N x 32 bytes.  INTEGRATION.md shows the cv::Mat-typed variant a
// pilotguru maintainer drops into Frame::ExtractORB.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pgorb.h"

namespace pgorb {

struct Image8 {                       // CV_8UC1 view
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0, step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

class ORBextractor {
 public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int maxWidth = 1920, int maxHeight = 1080, int maxBatch = 1, int device = 0)
        : nlevels_(nlevels), scaleFactor_(scaleFactor)
    {
        pgorb_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxWidth, maxHeight, maxBatch, device, 0};
        if (pgorb_create(&p, &ctx_) != PGORB_OK) throw std::runtime_error(pgorb_last_error(nullptr));
        const int n = nlevels + 1;
        mvScaleFactor.resize(n); mvInvScaleFactor.resize(n); mvLevelSigma2.resize(n); mvInvLevelSigma2.resize(n);
        pgorb_scale_tables(ctx_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
    }
    ~ORBextractor() { pgorb_destroy(ctx_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // operator()(image, mask, keypoints, descriptors); the mask is ignored as in the reference
    // (ORBextractor.h:58); an empty image returns silently with outputs untouched (:1045).
    void operator()(const Image8& image, const Image8& /*mask*/, std::vector<pgorb_keypoint>& keypoints,
                    std::vector<uint8_t>& descriptors)
    {
        if (image.empty()) return;
        const int cap = pgorb_max_keypoints(ctx_, image.cols, image.rows);
        if (cap < 0) throw std::runtime_error("frame size unusable for the ORB cell grid");
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        if (pgorb_extract(ctx_, image.data, image.cols, image.rows, image.step, keypoints.data(), descriptors.data(), cap, &n) != PGORB_OK)
            throw std::runtime_error(pgorb_last_error(ctx_));
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
    pgorb_ctx* context() { return ctx_; }

 private:
    pgorb_ctx* ctx_ = nullptr;
    int nlevels_;
    float scaleFactor_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// The slice of ORB_SLAM2::Frame the matcher needs (keypoints == undistorted keypoints, k1 == 0).
struct Frame {
    std::vector<pgorb_keypoint> mvKeysUndistorted;
    std::vector<uint8_t> mDescriptors;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;      // ComputeImageBounds, Frame.cc:461-466
    int N() const { return (int)mvKeysUndistorted.size(); }
};

class ORBmatcher {
 public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;     // ORBmatcher.cc:38-40
    ORBmatcher(pgorb_ctx* ctx, float nnratio = 0.6f, bool checkOri = true)
        : ctx_(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return pgorb_descriptor_distance(a, b); }
    // vbPrevMatched: 2 floats per F1 keypoint, updated in place; vnMatches12 resized to F1.N()
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<float>& vbPrevMatched,
                                std::vector<int32_t>& vnMatches12, int windowSize = 10)
    {
        vnMatches12.assign(F1.N(), -1);
        if (F1.N() == 0) return 0;
        const int rc = pgorb_search_for_initialization(ctx_, F1.mvKeysUndistorted.data(), F1.mDescriptors.data(), F1.N(),
                                                       F2.mvKeysUndistorted.data(), F2.mDescriptors.data(), F2.N(),
                                                       F2.mnMinX, F2.mnMaxX, F2.mnMinY, F2.mnMaxY,
                                                       vbPrevMatched.data(), vnMatches12.data(), windowSize, mfNNratio,
                                                       mbCheckOrientation ? 1 : 0);
        if (rc < 0) throw std::runtime_error(pgorb_last_error(ctx_));
        return rc;
    }

 private:
    pgorb_ctx* ctx_;
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace pgorb
