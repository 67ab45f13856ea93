/*
 * amos_host_types.h -- plain-data views of the Frame / MapPoint members that the reference's
 * ORBmatcher::Search* loops read (include/Frame.h, include/MapPoint.h).  The host-side ORBmatcher
 * (amos-slam_amd/host/ORBmatcher.h) and the test oracle take these instead of Frame&, so that the
 * matcher can be exercised without the rest of ORB-SLAM2; INTEGRATION.md shows how Tracking.cc
 * fills them from a Frame.
 */
#ifndef AMOS_HOST_TYPES_H
#define AMOS_HOST_TYPES_H

#include <stdint.h>
#include "amos_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* AMOS_FRAME_GRID_ROWS (48) and AMOS_FRAME_GRID_COLS (64) come from amos_frontend.h (Frame.h:56,61). */

/* What a search reads of the frame it searches IN (Frame::mvKeysUn, mDescriptors, mvuRight, the
 * undistorted image bounds mnMinX.. and the 64x48 feature grid built from them). */
typedef struct amos_frame_view {
    int32_t n;                     /* Frame::N */
    const amos_keypoint *keys_un;  /* Frame::mvKeysUn */
    const uint8_t *descriptors;    /* Frame::mDescriptors, n x 32 */
    const float *u_right;          /* Frame::mvuRight, or NULL (monocular: all -1) */
    float min_x, max_x, min_y, max_y; /* Frame::mnMinX, mnMaxX, mnMinY, mnMaxY */
} amos_frame_view;

/* One map point of the last frame projected into the current one: the values
 * ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) computes per point before its
 * candidate loop (ORBmatcher.cc:1613-1642). */
typedef struct amos_proj_query {
    float u, v;        /* projection */
    float invz;        /* invzc, used for the right-coordinate gate */
    int32_t octave;    /* LastFrame.mvKeys[i].octave */
    float angle;       /* LastFrame.mvKeysUn[i].angle */
    int32_t has_obs;   /* pMP->Observations() > 0 */
    uint8_t desc[32];  /* pMP->GetDescriptor() */
} amos_proj_query;

/* One local map point as ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) sees it
 * (ORBmatcher.cc:83-103): already filtered for mbTrackInView && !isBad(). */
typedef struct amos_map_query {
    float proj_x, proj_y, proj_xr; /* mTrackProjX, mTrackProjY, mTrackProjXR */
    float view_cos;                /* mTrackViewCos */
    int32_t level;                 /* mnTrackScaleLevel */
    int32_t has_obs;               /* Observations() > 0 (matters once it is assigned to a feature) */
    uint8_t desc[32];
} amos_map_query;

/* One map point of a keyframe projected into the current frame by
 * ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) -- the
 * relocalisation search (ORBmatcher.cc:1731-1863): already filtered for !isBad(), !sAlreadyFound, image bounds
 * and the distance-invariance range (:1758-1790). */
typedef struct amos_kf_query {
    float u, v;        /* projection (:1769-1770) */
    int32_t level;     /* nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame) */
    float angle;       /* pKF->mvKeysUn[i].angle */
    uint8_t desc[32];  /* pMP->GetDescriptor() */
} amos_kf_query;
/* One map point projected into a KEYFRAME by the mapping / loop-closing searches -- Fuse (ORBmatcher.cc:1020-1177,
 * 1179-1312), SearchByProjection(pKF, Scw, ...) (:388-512), SearchBySim3 (:1314-1565) -- after the caller's geometric
 * filtering (depth, image bounds, distance invariance, viewing angle). */
typedef struct amos_window_query {
    float u, v;        /* projection */
    float ur;          /* Fuse #1 only: u - bf * invz (:1064) */
    int32_t level;     /* nPredictedLevel */
    int32_t src;       /* SearchBySim3: index of the source feature (i1 or i2); free otherwise */
    uint8_t desc[32];  /* pMP->GetDescriptor() */
} amos_window_query;
#define AMOS_MATCH_FREE (-1)   /* CurrentFrame.mvpMapPoints[i2] == NULL */
#define AMOS_MATCH_TAKEN (-2)  /* occupied on entry by a map point that is not one of the queries */

/* What ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches) reads of either side
 * (ORBmatcher.cc:230-382): descriptors, keypoint angles, the DBoW2::FeatureVector (node id -> feature indices,
 * node ids ascending as std::map iterates them) and, for the keyframe, which features carry a good map point.
 * The keyframe-keyframe searches (SearchByBoW(KF,KF) :656-808, SearchForTriangulation :810-1018) read the same. */
typedef struct amos_bow_view {
    int32_t n;                    /* features */
    const amos_keypoint *keys;    /* KF: mvKeysUn, F: mvKeys (only .angle is read) */
    const uint8_t *descriptors;   /* n x 32 */
    const uint8_t *has_point;     /* KF: vpMapPointsKF[i] && !isBad(); NULL = all; ignored for F.  SearchForTriangulation:
                                     GetMapPoint(i) != NULL (those features are skipped) */
    const float *u_right;         /* mvuRight (SearchForTriangulation only), or NULL: monocular, all -1 */
    int32_t n_nodes;
    const uint32_t *node_ids;     /* ascending */
    const int32_t *node_off;      /* n_nodes + 1 */
    const int32_t *node_idx;      /* feature indices, in the FeatureVector's order */
} amos_bow_view;

#ifdef __cplusplus
}
#endif
#endif
