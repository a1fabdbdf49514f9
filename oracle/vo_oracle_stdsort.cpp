// CPU oracle, TEST INFRASTRUCTURE ONLY (see vo_oracle.h): the reference's top-N
// truncation done literally.
//
// reference src/feature_detector.cpp:62-68 keeps the max_features strongest
// FAST corners with std::sort (unstable) under the comparator
// `a.response > b.response` and a resize().  FAST scores are small integers, so
// many corners tie at the cut and which of them survive -- and the order of
// equal-response neighbours everywhere in the list -- is whatever the C++
// library's introsort leaves.  This file runs exactly that: the same algorithm
// call, the same comparator, the same input order (cv::FAST emits raster order),
// compiled with the libstdc++ of gcc 11 (the one Ubuntu 22.04 / ROS 2 Humble
// builds the reference with).  The sequence of comparisons and swaps of
// std::sort depends only on the comparison results, not on the element type, so
// the 24-byte record used here orders like the reference's cv::KeyPoint.
//
// The canonical selection (ora_select_top: response descending, raster index
// ascending) is what the GPU path reproduces bit for bit; this function exists
// to measure how far that canonical choice is from the literal behaviour.
#include <algorithm>
#include <vector>

#include "vo_oracle.h"

extern "C" int ora_select_top_stdsort(ora_kp *kps, int n, int max_features)
{
    if (n <= max_features) return n;  // reference :62: left untouched, raster order
    std::vector<ora_kp> v(kps, kps + n);
    std::sort(v.begin(), v.end(), [](const ora_kp &a, const ora_kp &b) { return a.response > b.response; });
    v.resize((size_t)max_features);
    std::copy(v.begin(), v.end(), kps);
    return max_features;
}
