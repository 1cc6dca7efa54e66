// slslam_amd/host/window_packer.cpp — see window_packer.h.
#include "window_packer.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

extern "C" int slslam_pack_window(const slslam_keyframe* kfs, int num_kfs, const slslam_landmark* lms, int num_lms,
                                  int W, slslam_packed_window* out) {
  if (!out || num_kfs < 0 || num_lms < 0 || W < 0) return 1;
  std::memset(out, 0, sizeof(*out));
  std::map<int, const slslam_keyframe*> kf_by_id;      // kfs / ba_kfs are std::map in the reference: ascending id
  for (int i = 0; i < num_kfs; ++i) kf_by_id[kfs[i].id] = &kfs[i];
  std::map<int, const slslam_landmark*> lm_by_id;
  for (int i = 0; i < num_lms; ++i) lm_by_id[lms[i].id] = &lms[i];

  std::vector<int> camera_index, fixed_index, line_index, cam_kf, line_lm;
  std::vector<const double*> obs_ptr;
  std::vector<double> cam_param, line_param;
  std::map<int, int> kfid_map, lm_count;
  int kfidx = 0;
  for (auto& kv : kf_by_id) {                          // slam.cpp:811-832
    const slslam_keyframe* kf = kv.second;
    if (kf->ba_rank < 0 || kf->ba_rank >= W) continue;
    for (int m = 0; m < kf->num_member_lms; ++m) lm_count[kf->member_lms[m]]++;
    double wt[6];
    slslam_gc_Rt_to_wt(&kf->T, wt);
    cam_param.insert(cam_param.end(), wt, wt + 6);
    cam_kf.push_back(kf->id);
    kfid_map[kf->id] = kfidx++;
  }
  int lmidx = 0;
  for (auto& lc : lm_count) {                          // slam.cpp:838-888
    if (lc.second < 2) continue;
    auto lit = lm_by_id.find(lc.first);
    if (lit == lm_by_id.end()) continue;
    const slslam_landmark* lm = lit->second;
    bool placed = false;
    for (int j = 0; j < lm->num_obs; ++j) {
      const slslam_observation& ob = lm->obs[j];
      auto kit = kf_by_id.find(ob.kf_id);
      if (kit == kf_by_id.end() || kit->second->ba_rank < 0) continue;      // not in ba_kfs
      auto iit = kfid_map.find(ob.kf_id);
      if (iit == kfid_map.end()) {                     // a keyframe of rank >= W: constant camera, appended
        fixed_index.push_back(1);
        camera_index.push_back(kfidx);
        double wt[6];
        slslam_gc_Rt_to_wt(&kit->second->T, wt);
        cam_param.insert(cam_param.end(), wt, wt + 6);
        cam_kf.push_back(ob.kf_id);
        kfid_map[ob.kf_id] = kfidx++;
      } else {
        fixed_index.push_back(iit->second < W ? 0 : 1);
        camera_index.push_back(iit->second);
      }
      line_index.push_back(lmidx);
      obs_ptr.push_back(ob.obs);
      placed = true;
    }
    (void)placed;
    auto ikf = kf_by_id.find(lm->init_kf_id);
    if (ikf == kf_by_id.end()) return 1;
    double line_w[6], orth[4];
    slslam_gc_line_from_pose(lm->line, &ikf->second->T, line_w);            // :884-886
    slslam_gc_av_to_orth(line_w, orth);
    line_param.insert(line_param.end(), orth, orth + 4);
    line_lm.push_back(lm->id);
    ++lmidx;
  }
  const int C = (int)cam_kf.size(), L = (int)line_lm.size(), M = (int)camera_index.size();
  out->num_cameras = C; out->num_lines = L; out->num_observations = M; out->num_parameters = 6 * C + 4 * L;
  out->camera_index = new int[M > 0 ? M : 1];
  out->line_index = new int[M > 0 ? M : 1];
  out->fixed_index = new int[M > 0 ? 2 * M : 1];
  out->observations = new double[M > 0 ? 8 * M : 1];
  out->parameters = new double[out->num_parameters > 0 ? out->num_parameters : 1];
  out->camera_kf_id = new int[C > 0 ? C : 1];
  out->line_lm_id = new int[L > 0 ? L : 1];
  for (int i = 0; i < M; ++i) {                        // slam.cpp:905-912
    out->camera_index[i] = camera_index[i];
    out->line_index[i] = line_index[i];
    out->fixed_index[2 * i] = fixed_index[i];
    out->fixed_index[2 * i + 1] = 0;
    std::memcpy(out->observations + 8 * (size_t)i, obs_ptr[i], 8 * sizeof(double));
  }
  if (C) std::memcpy(out->parameters, cam_param.data(), sizeof(double) * 6 * C);
  if (L) std::memcpy(out->parameters + 6 * (size_t)C, line_param.data(), sizeof(double) * 4 * L);
  if (C) std::memcpy(out->camera_kf_id, cam_kf.data(), sizeof(int) * C);
  if (L) std::memcpy(out->line_lm_id, line_lm.data(), sizeof(int) * L);
  return 0;
}

extern "C" int slslam_unpack_window(const slslam_packed_window* w, slslam_keyframe* kfs, int num_kfs,
                                    slslam_landmark* lms, int num_lms) {
  if (!w) return 1;
  std::map<int, slslam_keyframe*> kf_by_id;
  for (int i = 0; i < num_kfs; ++i) kf_by_id[kfs[i].id] = &kfs[i];
  std::map<int, slslam_landmark*> lm_by_id;
  for (int i = 0; i < num_lms; ++i) lm_by_id[lms[i].id] = &lms[i];
  for (int c = 0; c < w->num_cameras; ++c) {           // slam.cpp:957-962
    auto it = kf_by_id.find(w->camera_kf_id[c]);
    if (it == kf_by_id.end()) return 1;
    slslam_gc_wt_to_Rt(w->parameters + 6 * (size_t)c, &it->second->T);
  }
  for (int l = 0; l < w->num_lines; ++l) {             // slam.cpp:964-972
    auto it = lm_by_id.find(w->line_lm_id[l]);
    if (it == lm_by_id.end()) return 1;
    auto kit = kf_by_id.find(it->second->init_kf_id);
    if (kit == kf_by_id.end()) return 1;
    double line_w[6];
    slslam_gc_orth_to_av(w->parameters + 6 * (size_t)w->num_cameras + 4 * (size_t)l, line_w);
    slslam_gc_line_to_pose(line_w, &kit->second->T, it->second->line);
  }
  return 0;
}

extern "C" void slslam_free_packed_window(slslam_packed_window* w) {
  if (!w) return;
  delete[] w->camera_index; delete[] w->line_index; delete[] w->fixed_index;
  delete[] w->observations; delete[] w->parameters; delete[] w->camera_kf_id; delete[] w->line_lm_id;
  std::memset(w, 0, sizeof(*w));
}

// ---- pose graph (SLAM::pose_optimization, reference src/slam.cpp:1236-1313)
extern "C" int slslam_pack_pose_graph(const slslam_pose* kf_T, int num_poses, slslam_pg_edge* edges, int num_edges,
                                      slslam_packed_pose_graph* out) {
  if (!out || num_poses < 0 || num_edges < 0 || (num_poses > 0 && !kf_T) || (num_edges > 0 && !edges)) return 1;
  std::memset(out, 0, sizeof(*out));
  for (int i = 0; i < num_edges; ++i)
    if (edges[i].n1 < 0 || edges[i].n1 >= num_poses || edges[i].n2 < 0 || edges[i].n2 >= num_poses) return 1;
  std::stable_sort(edges, edges + num_edges, [](const slslam_pg_edge& a, const slslam_pg_edge& b) {
    return a.n1 != b.n1 ? a.n1 < b.n1 : a.n2 < b.n2;                 // std::set<pii> iteration order (slam.cpp:1249)
  });
  for (int i = 1; i < num_edges; ++i)
    if (edges[i].n1 == edges[i - 1].n1 && edges[i].n2 == edges[i - 1].n2) return 1;   // a set holds a pair once
  out->num_poses = num_poses; out->num_edges = num_edges;
  out->pose_index_1 = new int[num_edges > 0 ? num_edges : 1];
  out->pose_index_2 = new int[num_edges > 0 ? num_edges : 1];
  out->constraints = new double[6 * (size_t)(num_edges > 0 ? num_edges : 1)];
  out->parameters = new double[6 * (size_t)(num_poses > 0 ? num_poses : 1)];
  for (int i = 0; i < num_edges; ++i) {                                // slam.cpp:1267-1274
    out->pose_index_1[i] = edges[i].n1;
    out->pose_index_2[i] = edges[i].n2;
    slslam_gc_Rt_to_wt(&edges[i].C, out->constraints + 6 * (size_t)i);
  }
  for (int k = 0; k < num_poses; ++k) slslam_gc_Rt_to_wt(&kf_T[k], out->parameters + 6 * (size_t)k);   // :1276-1280
  return 0;
}

extern "C" int slslam_unpack_pose_graph(const slslam_packed_pose_graph* g, slslam_pose* kf_T, int num_poses,
                                        slslam_pg_edge* edges, int num_edges) {
  if (!g || g->num_poses != num_poses || (num_poses > 0 && !kf_T) || (num_edges > 0 && !edges)) return 1;
  for (int k = 0; k < num_poses; ++k) slslam_gc_wt_to_Rt(g->parameters + 6 * (size_t)k, &kf_T[k]);    // slam.cpp:1295-1300
  for (int i = 0; i < num_edges; ++i) {                                                               // :1302-1310
    if (edges[i].n1 < 0 || edges[i].n1 >= num_poses || edges[i].n2 < 0 || edges[i].n2 >= num_poses) return 1;
    slslam_gc_T_21(&kf_T[edges[i].n2], &kf_T[edges[i].n1], &edges[i].T);
    slslam_gc_T_21(&kf_T[edges[i].n1], &kf_T[edges[i].n2], &edges[i].T_rev);
  }
  return 0;
}

extern "C" void slslam_free_packed_pose_graph(slslam_packed_pose_graph* g) {
  if (!g) return;
  delete[] g->pose_index_1; delete[] g->pose_index_2; delete[] g->constraints; delete[] g->parameters;
  std::memset(g, 0, sizeof(*g));
}

// ---- motion-only bundle adjustment (SLAM::motion_only_ba, reference src/slam.cpp:578-675)
extern "C" int slslam_pack_motion_only(const slslam_pose* T, const double* obs_cur, const double* obs_prev, const double* lines,
                                       int num_inliers, slslam_packed_window* out) {
  if (!T || !out || num_inliers < 0 || (num_inliers > 0 && (!obs_cur || !obs_prev || !lines))) return 1;
  std::memset(out, 0, sizeof(*out));
  const int K = num_inliers, M = 2 * K;
  out->num_cameras = 2; out->num_lines = K; out->num_observations = M; out->num_parameters = 12 + 4 * K;
  out->camera_index = new int[M > 0 ? M : 1];
  out->line_index = new int[M > 0 ? M : 1];
  out->fixed_index = new int[2 * (size_t)(M > 0 ? M : 1)];
  out->observations = new double[8 * (size_t)(M > 0 ? M : 1)];
  out->parameters = new double[out->num_parameters];
  slslam_gc_Rt_to_wt(T, out->parameters);                              // camera 0: the current estimate (slam.cpp:590)
  slslam_pose I;
  std::memset(&I, 0, sizeof(I));
  I.R[0] = I.R[4] = I.R[8] = 1.0;
  slslam_gc_Rt_to_wt(&I, out->parameters + 6);                        // camera 1: pose_t() (:591)
  for (int i = 0; i < K; ++i) {                                        // :593-607
    out->camera_index[2 * i] = 0; out->line_index[2 * i] = i;
    out->fixed_index[4 * i] = 0; out->fixed_index[4 * i + 1] = 1;
    std::memcpy(out->observations + 16 * (size_t)i, obs_cur + 8 * (size_t)i, 8 * sizeof(double));
    out->camera_index[2 * i + 1] = 1; out->line_index[2 * i + 1] = i;
    out->fixed_index[4 * i + 2] = 1; out->fixed_index[4 * i + 3] = 1;
    std::memcpy(out->observations + 16 * (size_t)i + 8, obs_prev + 8 * (size_t)i, 8 * sizeof(double));
    slslam_gc_av_to_orth(lines + 6 * (size_t)i, out->parameters + 12 + 4 * (size_t)i);
  }
  return 0;
}

extern "C" void slslam_unpack_motion_only(const slslam_packed_window* w, slslam_pose* T) {
  if (w && T && w->parameters) slslam_gc_wt_to_Rt(w->parameters, T);  // slam.cpp:668-674
}
