// compiled with -DAFV_WITH_OPENCV -fsyntax-only against tests/opencv_mock: every template of the adapter that touches
// cv::Mat / cv::KeyPoint is instantiated the way INTEGRATION.md's subclasses do.
#include "afv_adapter.hpp"

struct Image {  // include/Image.h: the extractor reads grayImg
    cv::Mat grayImg;
};

void instantiate(afv::FeatureExtractor_orb32_hip &orb, afv::FeatureExtractor_akaze61_hip &akz, const Image &im) {
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    std::vector<afv::Mat2f> s2, inf;
    std::vector<float> size;
    orb(im, kps, desc, s2, inf, size);
    orb(im, kps, desc);
    orb.detectAndCompute(im, kps, desc);
    akz.detectAndCompute(im, kps, desc);
    std::vector<cv::Mat> pyramid;  // mvImagePyramid, read by the stereo matcher (Frame.cc:475)
    orb.ImagePyramid(pyramid);
    (void)orb.GetKeypointSize(kps[0]);
    (void)akz.GetKeypointOctave(kps[0]);
}
