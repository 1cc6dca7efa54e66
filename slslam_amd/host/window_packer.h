// slslam_amd/host/window_packer.h — the array packer around the LBA hot path (SURVEY.md 8a row 7,
// 8f rank 2): what SLAM::bundle_adjustment does before and after ceres::Solve, on plain structs.
//   pack    reference src/slam.cpp:811-921   map -> camera_index / line_index / fixed_index / observations / parameters
//   unpack  reference src/slam.cpp:957-972   parameters -> keyframe poses and landmark lines
// Ordering rules kept from the reference (its containers are std::map, i.e. ascending ids):
//   * free cameras = keyframes of ba_kfs with rank < W, in ascending keyframe id (:811-832);
//   * a landmark enters iff it is a member of >= 2 free keyframes (:838-840), landmarks in ascending id;
//   * of each landmark, every observation whose keyframe is in ba_kfs, in obs_vec order (:848-882);
//     keyframes of rank >= W are appended as constant cameras when first met (:855-864);
//   * line parameter = gc_av_to_orth(gc_line_from_pose(lm->line, init_kf->T)) (:884-886); lines are never
//     constant (fixed_index[2i+1] = 0, :908-909);
//   * write-back: every camera incl. the constant ones (:957-962), lines back into the init keyframe's frame
//     with its UPDATED pose (:964-972).
// The arrays are allocated with new[] so that ceres::LBAProblem can take ownership as in the reference.
#ifndef SLSLAM_WINDOW_PACKER_H_
#define SLSLAM_WINDOW_PACKER_H_

#include "gc_lite.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slslam_keyframe {
  int id;
  int ba_rank;                 /* ba_kfs[id] (rank by path length from the newest keyframe); < 0: not in ba_kfs */
  slslam_pose T;               /* world -> camera */
  const int* member_lms;       /* landmark ids tracked in this keyframe (keyframe_t::member_lms) */
  int num_member_lms;
} slslam_keyframe;

typedef struct slslam_observation { int kf_id; double obs[8]; } slslam_observation;   /* obs_t */

typedef struct slslam_landmark {
  int id;
  double line[6];              /* (closest point, direction) in the frame of keyframe init_kf_id */
  int init_kf_id;
  const slslam_observation* obs;
  int num_obs;
} slslam_landmark;

typedef struct slslam_packed_window {
  int num_cameras, num_lines, num_observations, num_parameters;
  int* camera_index;           /* new[]-allocated, as the reference allocates them (slam.cpp:899-903) */
  int* line_index;
  int* fixed_index;
  double* observations;
  double* parameters;
  int* camera_kf_id;           /* [C] keyframe id of each camera slot (vec_kfs) */
  int* line_lm_id;             /* [L] landmark id of each line slot (vec_lms)  */
} slslam_packed_window;

/* Returns 0, or 1 on inconsistent input (unknown keyframe / landmark ids). */
int slslam_pack_window(const slslam_keyframe* kfs, int num_kfs, const slslam_landmark* lms, int num_lms,
                       int window_size, slslam_packed_window* out);
/* Writes the solved parameters back: kfs[].T and lms[].line are updated in place. */
int slslam_unpack_window(const slslam_packed_window* w, slslam_keyframe* kfs, int num_kfs,
                         slslam_landmark* lms, int num_lms);
/* Frees whatever pack allocated and the caller has not handed to an LBAProblem (pass NULL-ed pointers otherwise). */
void slslam_free_packed_window(slslam_packed_window* w);

/* ---- motion-only bundle adjustment: what SLAM::motion_only_ba does before and after ceres::Solve
 *   pack    reference src/slam.cpp:590-640   camera 0 = gc_Rt_to_wt(T) (free), camera 1 = identity (constant); per inlier
 *           two observations of one line - (camera 0, current frame's observation), then (camera 1, previous frame's) -,
 *           every line constant (fixed_index = {0,1},{1,1}), line parameter = gc_av_to_orth(line) with the line
 *           given in the previous frame's coordinates
 *   unpack  reference src/slam.cpp:668-674   T = gc_wt_to_Rt(parameters[0..5])
 * obs_cur / obs_prev: [8 K] the inliers' observations (x0 y0 x1 y1 x2 y2 x3 y3, normalised), lines: [6 K] (closest point,
 * direction).  out->camera_kf_id / line_lm_id stay NULL. */
int slslam_pack_motion_only(const slslam_pose* T, const double* obs_cur, const double* obs_prev, const double* lines,
                            int num_inliers, slslam_packed_window* out);
void slslam_unpack_motion_only(const slslam_packed_window* w, slslam_pose* T);

/* ---- pose graph: what SLAM::pose_optimization does before and after ceres::Solve
 *   pack    reference src/slam.cpp:1248-1280   edge_set (std::set<pii>: ascending (n1, n2)) -> pose_index_1 / pose_index_2,
 *           constraints[6 i] = gc_Rt_to_wt(edges[(n1, n2)].C), parameters[6 k] = gc_Rt_to_wt(kfs[k]->T) for k = 0..N-1
 *   unpack  reference src/slam.cpp:1295-1311   kfs[k]->T = gc_wt_to_Rt(parameters[6 k]); every edge's current relative
 *           pose T = gc_T_21(kfs[n2]->T, kfs[n1]->T) refreshed in both directions
 * Keyframe ids are the pose slots (kfs is indexed 0..N-1 in the reference).  Edge 0's first pose is the gauge
 * (po_problem.cpp:62-63).  Arrays are new[]-allocated so that ceres::POProblem can take ownership. */
typedef struct slslam_pg_edge {
  int n1, n2;                  /* n1 < n2 is not required; the pack sorts by (n1, n2) like std::set<pii> */
  slslam_pose C;               /* measured T_{n2 <- n1} */
  slslam_pose T;               /* current T_{n2 <- n1} from the keyframe poses (refreshed by unpack) */
  slslam_pose T_rev;           /* current T_{n1 <- n2} (edges[(n2, n1)].T in the reference) */
} slslam_pg_edge;

typedef struct slslam_packed_pose_graph {
  int num_poses, num_edges;
  int* pose_index_1;
  int* pose_index_2;
  double* constraints;         /* [6 E] */
  double* parameters;          /* [6 N] */
} slslam_packed_pose_graph;

/* Sorts `edges` in place into the std::set<pii> order.  Returns 0, or 1 on an edge that names an unknown pose or a duplicate. */
int slslam_pack_pose_graph(const slslam_pose* kf_T, int num_poses, slslam_pg_edge* edges, int num_edges,
                           slslam_packed_pose_graph* out);
/* Writes the solved poses back into kf_T[] and refreshes T / T_rev of every edge. */
int slslam_unpack_pose_graph(const slslam_packed_pose_graph* g, slslam_pose* kf_T, int num_poses,
                             slslam_pg_edge* edges, int num_edges);
void slslam_free_packed_pose_graph(slslam_packed_pose_graph* g);

#ifdef __cplusplus
}
#endif
#endif
