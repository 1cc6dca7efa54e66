// slslam_amd/host/gc_lite.h — the boundary encodings the reference's packers use around the hot
// path, restated on plain arrays (no Eigen):
//   pose  <-> (angle-axis, translation)   gc_Rt_to_wt / gc_wt_to_Rt / gc_Rodriguez   reference src/gc.cpp:24-49,173-184
//   SE(3) compose / invert                gc_T_inv / gc_T_20 / gc_T_21               reference src/gc.cpp:51-53,163-171
//   line  world <-> keyframe              gc_line_to_pose / gc_line_from_pose        reference src/gc.cpp:63-81
//   line  (closest point, direction) <-> orthonormal 4-vector
//                                         gc_av_to_orth / gc_orth_to_av              reference src/gc.cpp:361-379,419-442
// Convention (reference src/gc.cpp:55-57): p_camera = R p_world + t; R is row-major here.
// Host-side helpers for callers that hold the reference's map structures; not on the GPU path.
#ifndef SLSLAM_GC_LITE_H_
#define SLSLAM_GC_LITE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slslam_pose { double R[9]; double t[3]; } slslam_pose;   /* pose_t (reference src/all.h) */

void slslam_gc_rodrigues_to_R(const double w[3], double R[9]);          /* gc_Rodriguez(Vector3d)  */
void slslam_gc_R_to_rodrigues(const double R[9], double w[3]);          /* gc_Rodriguez(Matrix3d)  */
void slslam_gc_wt_to_Rt(const double wt[6], slslam_pose* T);            /* gc_wt_to_Rt             */
void slslam_gc_Rt_to_wt(const slslam_pose* T, double wt[6]);            /* gc_Rt_to_wt             */
void slslam_gc_T_inv(const slslam_pose* T, slslam_pose* Ti);            /* gc_T_inv                */
void slslam_gc_T_20(const slslam_pose* T21, const slslam_pose* T10, slslam_pose* T20);   /* T20 = T21 * T10 */
void slslam_gc_T_21(const slslam_pose* T20, const slslam_pose* T10, slslam_pose* T21);   /* T21 = T20 * T10^-1 */
void slslam_gc_line_to_pose(const double line_w[6], const slslam_pose* T, double line_c[6]);
void slslam_gc_line_from_pose(const double line_c[6], const slslam_pose* T, double line_w[6]);
void slslam_gc_av_to_orth(const double av[6], double orth[4]);
void slslam_gc_orth_to_av(const double orth[4], double av[6]);

#ifdef __cplusplus
}
#endif
#endif
