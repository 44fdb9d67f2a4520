// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// CPU restatement of the reference matcher.  Citations are file:line under
// /root/reference.  Quirks of SURVEY App. C are kept on purpose.
#include "vo_match.hpp"
#include <climits>

namespace vo {

// DescriptorDistance: src/FeatureMatcher.cpp:710-726 (8 x u32 XOR + SWAR popcount)
int descriptorDistance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// destributeRightKeys: src/FeatureMatcher.cpp:728-752.  The reference indexes
// `indexes[pos]` without a bounds check; rows outside [0, imageHeight) would be
// undefined behaviour there and are skipped here (never hit: keypoints stay
// >= 19 level-pixels inside the image).
static void distributeRightKeys(const Extractor& feRight, const std::vector<KeyPoint>& rightKeys,
                                int imageHeight, std::vector<std::vector<int>>& indexes) {
    indexes.assign(imageHeight, {});
    int count = 0;
    for (const KeyPoint& kp : rightKeys) {
        const int yKey = cvRoundF(kp.y);
        const float r = 2.0f * feRight.scalePyramid[kp.octave];
        const int mn = cvFloorF((float)yKey - r);
        const int mx = cvCeilF((float)yKey + r);
        for (int pos = mn; pos <= mx; pos++)
            if (pos >= 0 && pos < imageHeight) indexes[pos].push_back(count);
        count++;
    }
}

// findStereoMatchesORB2R: src/FeatureMatcher.cpp:528-708
void findStereoMatchesORB2R(const Extractor& feLeft, const Extractor& feRight, const Rig& rig,
                            TrackedKeys& keys, StereoStats* stats) {
    std::vector<std::vector<int>> indexes;
    distributeRightKeys(feRight, keys.rightKeyPoints, rig.height, indexes);
    const size_t leftEnd = keys.keyPoints.size();
    keys.estimatedDepth.assign(leftEnd, -1.0f);
    keys.close.assign(leftEnd, 0);
    keys.rightIdxs.assign(leftEnd, -1);
    keys.leftIdxs.assign(keys.rightKeyPoints.size(), -1);
    const int thDist = 75, closeNumber = 40;          // include/FeatureMatcher.h:25,36
    const float minD = 0;
    const float maxD = (float)rig.fx;
    const int windowRadius = 5, windowMovementX = 5;
    std::vector<std::pair<float, int>> allDepths;
    std::vector<std::pair<int, int>> allDists2;
    for (size_t leftRow = 0; leftRow < leftEnd; leftRow++) {
        const KeyPoint& kl = keys.keyPoints[leftRow];
        const int yKey = cvRoundF(kl.y);
        const float uL = kl.y;            // quirk: the disparity window is tested on y (:557)
        const int octL = kl.octave;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 256, bestIdx = -1;
        if (yKey < 0 || yKey >= (int)indexes.size()) continue;   // UB in the reference
        const std::vector<int>& bucket = indexes[yKey];
        if (bucket.empty()) continue;
        for (int idx : bucket) {
            const float uR = keys.rightKeyPoints[idx].y;     // quirk (:578)
            const int octR = keys.rightKeyPoints[idx].octave;
            if (octR < octL - 1 || octR > octL + 1) continue;
            if (!(uR >= minU && uR <= maxU)) continue;
            const int dist = descriptorDistance(&keys.Desc[leftRow * 32], &keys.rightDesc[(size_t)idx * 32]);
            if (stats) stats->candidates++;
            if (bestDist > dist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist > thDist) continue;
        if (stats) stats->sadRefinements++;
        const KeyPoint& kR = keys.rightKeyPoints[bestIdx];
        const float kRx = kR.x;
        const float scale = feLeft.scaleInvPyramid[octL];
        const float scuL = std::round(kl.x * scale);
        const float scvL = std::round(kl.y * scale);
        const float scuR = std::round(kRx * scale);
        const Image& imL = feLeft.imagePyramid[octL];
        const Image& imR = feRight.imagePyramid[octL];
        const int ly0 = (int)(scvL - windowRadius), lx0 = (int)(scuL - windowRadius);
        int bestDistW = INT_MAX;
        int bestX = 0;
        float allDists[2 * 5 + 1];
        for (int i = 0; i < 11; i++) allDists[i] = 0.f;
        for (int xMov = -windowMovementX; xMov <= windowMovementX; xMov++) {
            const float startW = scuR + xMov - windowRadius;
            const float endW = scuR + xMov + windowRadius + 1;
            if (startW < 0 || endW >= imR.w) continue;
            const int rx0 = (int)startW;
            long long sad = 0;   // cv::norm(L1) on u8 = integer SAD (as double)
            for (int r = 0; r < 11; r++)
                for (int c = 0; c < 11; c++)
                    sad += std::abs((int)imL.at(ly0 + r, lx0 + c) - (int)imR.at(ly0 + r, rx0 + c));
            const float dist = (float)(double)sad;
            if ((float)bestDistW > dist) { bestX = xMov; bestDistW = (int)dist; }
            allDists[xMov + windowMovementX] = dist;
        }
        if (bestX == -windowMovementX || bestX == windowMovementX) continue;
        const float dist1 = allDists[windowMovementX + bestX - 1];
        const float dist2 = allDists[windowMovementX + bestX];
        const float dist3 = allDists[windowMovementX + bestX + 1];
        const float delta = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (delta > 1 || delta < -1) continue;
        if (stats) stats->matches++;
        const float newuR = feLeft.scalePyramid[octL] * ((float)scuR + (float)bestX + delta);
        const float disparity = kl.x - newuR;
        if (disparity > 0.0f && (double)disparity < rig.fx) {
            const float depth = ((float)rig.fx * rig.baseline) / disparity;
            keys.rightIdxs[leftRow] = bestIdx;
            keys.leftIdxs[bestIdx] = (int)leftRow;
            keys.estimatedDepth[leftRow] = depth;
            allDists2.emplace_back(bestDistW, (int)leftRow);
            allDepths.emplace_back(depth, (int)leftRow);
            if (depth < rig.baseline * closeNumber) keys.close[leftRow] = 1;
        }
    }
    if (allDepths.empty()) return;
    std::sort(allDepths.begin(), allDepths.end());
    std::sort(allDists2.begin(), allDists2.end());
    const int medianD = allDists2[allDists2.size() / 2].first;
    const float medDistD = (float)medianD * (1.5f * 1.4f);
    const int endDe = cvFloorD(allDepths.size() * 0.01);
    auto invalidate = [&](int l) {
        const int rIdx = keys.rightIdxs[l];
        if (rIdx >= 0) keys.leftIdxs[rIdx] = -1;
        keys.rightIdxs[l] = -1;
        keys.estimatedDepth[l] = -1;
        keys.close[l] = 0;
    };
    for (int i = 0; i < endDe; i++) invalidate(allDepths[i].second);
    for (int i = (int)allDists2.size() - 1; i >= 0; i--) {
        if ((float)allDists2[i].first < medDistD) break;
        invalidate(allDists2[i].second);
    }
}

// assignKeysToGrids: src/FeatureTracker.cpp:28-54
void assignKeysToGrids(TrackedKeys& keys, const std::vector<KeyPoint>& kps,
                       std::vector<std::vector<int>>& grid, int width, int height) {
    const float imageRatio = (float)width / (float)height;
    keys.xGrids = 64;
    keys.yGrids = cvCeilF((float)keys.xGrids / imageRatio);
    keys.xMult = (float)keys.xGrids / (float)width;
    keys.yMult = (float)keys.yGrids / (float)height;
    grid.assign((size_t)keys.yGrids * keys.xGrids, {});
    int count = 0;
    for (const KeyPoint& kp : kps) {
        int xPos = cvRoundF(kp.x * keys.xMult);
        int yPos = cvRoundF(kp.y * keys.yMult);
        if (xPos < 0) xPos = 0;
        if (yPos < 0) yPos = 0;
        if (xPos >= keys.xGrids) xPos = keys.xGrids - 1;
        if (yPos >= keys.yGrids) yPos = keys.yGrids - 1;
        grid[(size_t)yPos * keys.xGrids + xPos].push_back(count++);
    }
}

// getMatchIdxs: src/FeatureMatcher.cpp:13-64
void getMatchIdxs(float trackX, float trackY, std::vector<int>& idxs, const TrackedKeys& keys,
                  int predictedScale, float radius, bool right) {
    const int minX = std::max(0, cvFloorF((trackX - radius) * keys.xMult));
    const int maxX = std::min(keys.xGrids - 1, cvCeilF((trackX + radius) * keys.xMult));
    const int minY = std::max(0, cvFloorF((trackY - radius) * keys.yMult));
    const int maxY = std::min(keys.yGrids - 1, cvCeilF((trackY + radius) * keys.yMult));
    if (minX >= keys.xGrids || minY >= keys.yGrids || maxX < 0 || maxY < 0) return;
    const int maxLevel = predictedScale + 1, minLevel = predictedScale - 1;
    const std::vector<std::vector<int>>& G = right ? keys.rkeyGrid : keys.lkeyGrid;
    const std::vector<KeyPoint>& K = right ? keys.rightKeyPoints : keys.keyPoints;
    for (int row = minY; row <= maxY; row++)
        for (int col = minX; col <= maxX; col++) {
            const std::vector<int>& cell = G[(size_t)row * keys.xGrids + col];
            for (int gi : cell) {
                const KeyPoint& kp = K[gi];
                if (kp.octave > maxLevel || kp.octave < minLevel) continue;
                const float distx = kp.x - trackX, disty = kp.y - trackY;
                if (std::fabs(distx) < radius && std::fabs(disty) < radius) idxs.push_back(gi);
            }
        }
}

// matchByProjectionRPred: src/FeatureMatcher.cpp:254-389 (greedy, order-dependent claims)
int matchByProjectionRPred(const Extractor& feLeft, const std::vector<MapPointView>& mps,
                           const TrackedKeys& keys, std::vector<int>& matchedIdxsL,
                           std::vector<int>& matchedIdxsR, std::vector<std::pair<int, int>>& matchesIdxs,
                           float rad, long long* nCandidates) {
    const int matchDistProj = 100;     // include/FeatureMatcher.h:27
    const float ratioProj = 0.8f;      // :28
    int nMatches = 0;
    for (size_t i = 0; i < mps.size(); i++) {
        std::pair<int, int>& keyPair = matchesIdxs[i];
        const MapPointView& mp = mps[i];
        if (keyPair.first >= 0 || keyPair.second >= 0) continue;
        int predScaleLevel = mp.scaleLevelL;
        float radius = feLeft.scalePyramid[predScaleLevel] * rad;
        std::vector<int> idxs;
        getMatchIdxs(mp.predLx, mp.predLy, idxs, keys, predScaleLevel, radius, false);
        int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
        if (!idxs.empty() && mp.inFrame) {
            for (int idx : idxs) {
                if (matchedIdxsL[idx] >= 0) continue;
                const int lev = keys.keyPoints[idx].octave;
                const int dist = descriptorDistance(mp.desc, &keys.Desc[(size_t)idx * 32]);
                if (nCandidates) (*nCandidates)++;
                if (dist < bestDist) {
                    secDist = bestDist; bestLev2 = bestLev; bestDist = dist; bestLev = lev; bestIdx = idx;
                    continue;
                }
                if (dist < secDist) { secDist = dist; bestLev2 = lev; }
            }
        }
        std::vector<int> idxsR;
        predScaleLevel = mp.scaleLevelR;
        radius = feLeft.scalePyramid[predScaleLevel] * rad;
        getMatchIdxs(mp.predRx, mp.predRy, idxsR, keys, predScaleLevel, radius, true);
        int bestDistR = 256, bestIdxR = -1, bestLevR = -1, bestLevR2 = -1, secDistR = 256;
        if (!idxsR.empty() && mp.inFrameR) {
            for (int idx : idxsR) {
                if (matchedIdxsR[idx] >= 0) continue;
                const int lev = keys.rightKeyPoints[idx].octave;
                const int dist = descriptorDistance(mp.desc, &keys.rightDesc[(size_t)idx * 32]);
                if (nCandidates) (*nCandidates)++;
                if (dist < bestDistR) {
                    secDistR = bestDistR; bestLevR2 = bestLevR; bestDistR = dist; bestLevR = lev; bestIdxR = idx;
                    continue;
                }
                if (dist < secDistR) { secDistR = dist; bestLevR2 = lev; }
            }
        }
        bool right = false;
        if (bestDist > bestDistR) {
            bestDist = bestDistR; secDist = secDistR; bestLev = bestLevR; bestLev2 = bestLevR2;
            right = true;
        }
        if (bestDist > matchDistProj) continue;
        if (bestLev == bestLev2 && (float)bestDist >= ratioProj * (float)secDist) continue;
        if (bestLev != bestLev2 || (float)bestDist < ratioProj * (float)secDist) {
            nMatches++;
            if (right) {
                matchedIdxsR[bestIdxR] = (int)i;
                keyPair.second = bestIdxR;
                if (keys.leftIdxs[bestIdxR] >= 0) {
                    keyPair.first = keys.leftIdxs[bestIdxR];
                    matchedIdxsL[keys.leftIdxs[bestIdxR]] = (int)i;
                }
            } else {
                matchedIdxsL[bestIdx] = (int)i;
                keyPair.first = bestIdx;
                if (keys.rightIdxs[bestIdx] >= 0) {
                    keyPair.second = keys.rightIdxs[bestIdx];
                    matchedIdxsR[keys.rightIdxs[bestIdx]] = (int)i;
                }
            }
        }
    }
    return nMatches;
}

// matchByProjectionMono: src/FeatureMatcher.cpp:391-456
int matchByProjectionMono(const Extractor& feLeft, const std::vector<MapPointView>& mps, const TrackedKeys& keys,
                          std::vector<int>& matchedIdxsL, std::vector<std::pair<int, int>>& matchesIdxs, float rad,
                          long long* nCandidates) {
    const int matchDistProj = 100;     // include/FeatureMatcher.h:27
    const float ratioProj = 0.8f;      // :28
    int nMatches = 0;
    for (size_t i = 0; i < mps.size(); i++) {
        std::pair<int, int>& keyPair = matchesIdxs[i];
        const MapPointView& mp = mps[i];
        if (keyPair.first >= 0 || keyPair.second >= 0) continue;
        const int predScaleLevel = mp.scaleLevelL;
        const float radius = feLeft.scalePyramid[predScaleLevel] * rad;
        std::vector<int> idxs;
        getMatchIdxs(mp.predLx, mp.predLy, idxs, keys, predScaleLevel, radius, false);
        int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
        if (!idxs.empty() && mp.inFrame) {
            for (int idx : idxs) {
                if (matchedIdxsL[idx] >= 0) continue;
                const int lev = keys.keyPoints[idx].octave;
                const int dist = descriptorDistance(mp.desc, &keys.Desc[(size_t)idx * 32]);
                if (nCandidates) (*nCandidates)++;
                if (dist < bestDist) { secDist = bestDist; bestLev2 = bestLev; bestDist = dist; bestLev = lev; bestIdx = idx; continue; }
                if (dist < secDist) { secDist = dist; bestLev2 = lev; }
            }
        }
        if (bestDist > (matchDistProj + 50)) continue;
        if (bestLev == bestLev2 && bestDist >= (ratioProj + 0.1) * secDist) continue;
        if (bestLev != bestLev2 || bestDist < (ratioProj + 0.1) * secDist) {
            nMatches++;
            matchedIdxsL[bestIdx] = (int)i;
            keyPair.first = bestIdx;
        }
    }
    return nMatches;
}

// matchByRadius: src/FeatureMatcher.cpp:458-526
int matchByRadius(const Extractor& feLeft, const std::vector<KeyPoint>& lastKps, const std::vector<uint8_t>& lastDesc,
                  const TrackedKeys& actKeys, std::vector<int>& matchedIdxsL, float rad, std::vector<int>& matchOut) {
    const int matchDistProj = 100;
    const float ratioProj = 0.8f;
    const double pixelParallaxThresh = 10.0;
    int nMatches = 0;
    matchOut.assign(lastKps.size(), -1);
    for (size_t i = 0; i < lastKps.size(); i++) {
        const KeyPoint key = lastKps[i];
        const uint8_t* mpDesc = &lastDesc[i * 32];
        const int predScaleLevel = key.octave;
        const float radius = feLeft.scalePyramid[predScaleLevel] * rad;
        std::vector<int> idxs;
        getMatchIdxs(key.x, key.y, idxs, actKeys, predScaleLevel, radius, false);
        int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
        for (int idx : idxs) {
            if (matchedIdxsL[idx] >= 0) continue;
            const KeyPoint& kPL = actKeys.keyPoints[idx];
            const double dx = (double)kPL.x - (double)key.x, dy = (double)kPL.y - (double)key.y;
            if (!(std::sqrt(dx * dx + dy * dy) > pixelParallaxThresh)) continue;      // (p2 - p1).norm() > thresh
            const int lev = kPL.octave;
            const int dist = descriptorDistance(mpDesc, &actKeys.Desc[(size_t)idx * 32]);
            if (dist < bestDist) { secDist = bestDist; bestLev2 = bestLev; bestDist = dist; bestLev = lev; bestIdx = idx; continue; }
            if (dist < secDist) { secDist = dist; bestLev2 = lev; }
        }
        if (bestDist > matchDistProj) continue;
        if (bestLev == bestLev2 && (float)bestDist >= ratioProj * (float)secDist) continue;
        if (bestLev != bestLev2 || (float)bestDist < ratioProj * (float)secDist) {
            nMatches++;
            matchedIdxsL[bestIdx] = (int)i;
            matchOut[i] = bestIdx;
        }
    }
    return nMatches;
}

}  // namespace vo
