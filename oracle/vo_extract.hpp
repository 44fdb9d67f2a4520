// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates FeatureExtractor (reference include/FeatureExtractor.h:53-98,
// src/FeatureExtractor.cpp:268-682).
#pragma once
#include "vo_common.hpp"

namespace vo {

struct Extractor {
    // reference include/FeatureExtractor.h:62-80
    int nFeatures, nLevels;
    float imScale;
    int edgeThreshold, patchSize, halfPatchSize, maxFastThreshold, minFastThreshold;
    std::vector<int> scaledPatchSize, featurePerLevel, umax;
    std::vector<float> scalePyramid, scaleInvPyramid, sigmaFactor, InvSigmaFactor;
    std::vector<Image> imagePyramid;   // unblurred levels (border-less)
    std::vector<Image> blurPyramid;    // kept for tests (the reference blurs a temporary clone)
    std::vector<std::vector<KeyPoint>> fastCandidates;  // pre-SSC, kept for tests

    Extractor(int nfeatures = 2000, int nlevels = 8, float imscale = 1.2f, int edge = 19,
              int patch = 31, int maxFast = 20, int minFast = 7);

    void computePyramid(const Image& image);
    void computeKeypointsORBNew(std::vector<std::vector<KeyPoint>>& allKeys);
    std::vector<KeyPoint> ssc(std::vector<KeyPoint> keyPoints, int numRetPoints, float tolerance,
                              int cols, int rows) const;
    float computeOrientation(const Image& image, float px, float py) const;
    void extractKeysNew(const Image& image, std::vector<KeyPoint>& keypoints,
                        std::vector<uint8_t>& descriptors);
};

// third-party semantics restated [ext]
void resizeLinear8u(const Image& src, Image& dst);                      // cv::resize INTER_LINEAR
void fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold,
              std::vector<KeyPoint>& out);                              // cv::FAST(…, nms=true)
int fastCornerScore(const uint8_t* ptr, const int pixel[25], int threshold);
float fastAtan2(float y, float x);                                      // cv::fastAtan2
void gaussianKernel7Sigma2(int k[7]);                                   // 8.8 fixed-point taps
void gaussianBlur7(const Image& src, Image& dst);                       // cv::GaussianBlur 7x7 s=2
void orbDescriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc);

}  // namespace vo
