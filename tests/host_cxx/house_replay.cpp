// tests/host_cxx/house_replay.cpp — the caller SURVEY.md 8f-2 was written for: a program that holds the reference's map
// structures (keyframes with poses and member landmarks, landmarks with a line in their initial keyframe's frame and an
// observation list) and drives, per keyframe, the data flow of reference src/slam.cpp around its two solves with the host
// library and the C ABI - nothing else:
//   SLAM::motion_only_ba   (src/slam.cpp:578-675)   slslam_pack_motion_only -> slslam_lba_solve -> slslam_unpack_motion_only
//   SLAM::bundle_adjustment (src/slam.cpp:795-975)   slslam_pack_window      -> slslam_lba_solve -> slslam_unpack_window
//   SLAM::save_trajectory   (src/slam.cpp:1470-1496) slslam_write_trajectory after re-rooting at keyframe 0
// Keyframing, RANSAC and landmark management stay out of scope: the scene file (tools/house_study.py --dump-scene) supplies,
// per keyframe, the tracked observations, the stereo triangulation of every observed line and the visual odometry's motion.
//
//   house_replay <scene.bin> <poses_out.bin> <trajectory_out.txt>
// poses_out: frames x 12 doubles (R row-major, t) + 3 doubles (LM iterations, sum of initial costs, sum of final costs), then one
// record of 4 x uint64 per window solve: keyframe, cameras | lines << 16 | observations << 32, FNV-1a of the three index arrays,
// FNV-1a of the observation bytes - the part of a window that depends on the map's bookkeeping only, compared exactly.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../include/slslam_hip.h"
#include "gc_lite.h"
#include "sequence_io.h"
#include "window_packer.h"

namespace {
struct Landmark { double line[6]; int init_kf; std::vector<slslam_observation> obs; };
struct Frame { std::vector<int> ids; std::vector<double> obs, tri; double motion[6]; };

bool read_exact(FILE* f, void* p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; }
unsigned long long fnv1a(const void* p, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int solve(const slslam_packed_window& pk, int max_iter, slslam_summary* sum) {
  slslam_lba_window w;
  w.num_cameras = pk.num_cameras; w.num_lines = pk.num_lines; w.num_observations = pk.num_observations;
  w.camera_index = pk.camera_index; w.line_index = pk.line_index; w.fixed_index = pk.fixed_index;
  w.observations = pk.observations; w.parameters = pk.parameters;
  slslam_solver_options opt;
  slslam_default_options(&opt);
  opt.max_num_iterations = max_iter;
  return slslam_lba_solve(&w, &opt, sum, nullptr, 0, nullptr);
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: house_replay scene.bin poses_out.bin trajectory_out.txt\n"); return 64; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 66; }
  int hdr[3];
  if (!read_exact(f, hdr, sizeof(hdr))) return 65;
  const int frames = hdr[0], W = hdr[1], max_iter = hdr[2];
  std::vector<Frame> scene(frames);
  for (Frame& fr : scene) {
    int n = 0;
    if (!read_exact(f, &n, sizeof(n))) return 65;
    fr.ids.resize(n); fr.obs.resize(8 * (size_t)n); fr.tri.resize(6 * (size_t)n);
    if (!read_exact(f, fr.ids.data(), sizeof(int) * n) || !read_exact(f, fr.obs.data(), sizeof(double) * 8 * n) ||
        !read_exact(f, fr.tri.data(), sizeof(double) * 6 * n) || !read_exact(f, fr.motion, sizeof(fr.motion))) return 65;
  }
  std::fclose(f);

  std::vector<slslam_pose> kfT;                         // world -> camera, world = frame of keyframe 0
  std::vector<std::vector<int>> members;                // keyframe_t::member_lms
  std::map<int, Landmark> lms;                          // ascending id, as the reference's std::map
  double s_it = 0, s_c0 = 0, s_c1 = 0;
  std::vector<unsigned long long> digests;
  for (int k = 0; k < frames; ++k) {
    const Frame& fr = scene[k];
    slslam_pose T;
    std::memset(&T, 0, sizeof(T));
    T.R[0] = T.R[4] = T.R[8] = 1.0;
    if (k > 0) {
      // ---- pose_estimation: the odometry's relative motion, refined by motion-only BA against the previous keyframe
      slslam_pose motion;
      slslam_gc_wt_to_Rt(fr.motion, &motion);
      const Frame& pv = scene[k - 1];
      std::map<int, int> prev_pos;
      for (size_t i = 0; i < pv.ids.size(); ++i) prev_pos[pv.ids[i]] = (int)i;
      std::vector<double> obs_cur, obs_prev, lines;
      for (size_t i = 0; i < fr.ids.size(); ++i) {
        const int id = fr.ids[i];
        auto lit = lms.find(id);
        auto pit = prev_pos.find(id);
        if (lit == lms.end() || pit == prev_pos.end()) continue;
        obs_cur.insert(obs_cur.end(), fr.obs.begin() + 8 * i, fr.obs.begin() + 8 * i + 8);
        obs_prev.insert(obs_prev.end(), pv.obs.begin() + 8 * (size_t)pit->second, pv.obs.begin() + 8 * (size_t)pit->second + 8);
        double line_w[6], line_p[6];
        slslam_gc_line_from_pose(lit->second.line, &kfT[lit->second.init_kf], line_w);
        slslam_gc_line_to_pose(line_w, &kfT[k - 1], line_p);          // the line in the previous keyframe's frame
        lines.insert(lines.end(), line_p, line_p + 6);
      }
      const int K = (int)(lines.size() / 6);
      if (K >= 5) {
        slslam_packed_window pk;
        if (slslam_pack_motion_only(&motion, obs_cur.data(), obs_prev.data(), lines.data(), K, &pk)) return 70;
        slslam_summary sum;
        const int rc = solve(pk, max_iter, &sum);
        if (rc) { std::fprintf(stderr, "motion-only solve of keyframe %d: %s\n", k, slslam_status_string(rc)); return rc; }
        slslam_unpack_motion_only(&pk, &motion);
        slslam_free_packed_window(&pk);
      }
      slslam_gc_T_20(&motion, &kfT[k - 1], &T);
    }
    kfT.push_back(T);
    members.push_back(fr.ids);
    for (size_t i = 0; i < fr.ids.size(); ++i) {         // add_lms: new landmarks from the stereo triangulation, observation lists
      Landmark& lm = lms[fr.ids[i]];
      if (lm.obs.empty()) { std::memcpy(lm.line, fr.tri.data() + 6 * i, sizeof(lm.line)); lm.init_kf = k; }
      slslam_observation ob;
      ob.kf_id = k;
      std::memcpy(ob.obs, fr.obs.data() + 8 * i, sizeof(ob.obs));
      lm.obs.push_back(ob);
    }
    if (k == 0) continue;
    // ---- local_bundle_adjustment: ba_kfs = the 2 W newest keyframes ranked by distance from the newest (a chain: no loops)
    std::vector<slslam_keyframe> kfs(k + 1);
    for (int j = 0; j <= k; ++j) {
      kfs[j].id = j; kfs[j].ba_rank = (k - j < 2 * W) ? k - j : -1; kfs[j].T = kfT[j];
      kfs[j].member_lms = members[j].data(); kfs[j].num_member_lms = (int)members[j].size();
    }
    std::vector<slslam_landmark> lmv;
    lmv.reserve(lms.size());
    for (auto& kv : lms) {
      slslam_landmark l;
      l.id = kv.first; std::memcpy(l.line, kv.second.line, sizeof(l.line)); l.init_kf_id = kv.second.init_kf;
      l.obs = kv.second.obs.data(); l.num_obs = (int)kv.second.obs.size();
      lmv.push_back(l);
    }
    slslam_packed_window pk;
    if (slslam_pack_window(kfs.data(), (int)kfs.size(), lmv.data(), (int)lmv.size(), W, &pk)) return 71;
    if (pk.num_lines > 0) {
      unsigned long long h = fnv1a(pk.camera_index, sizeof(int) * pk.num_observations);
      h = fnv1a(pk.line_index, sizeof(int) * pk.num_observations, h);
      h = fnv1a(pk.fixed_index, sizeof(int) * 2 * pk.num_observations, h);
      digests.push_back((unsigned long long)k);
      digests.push_back((unsigned long long)pk.num_cameras | ((unsigned long long)pk.num_lines << 16) | ((unsigned long long)pk.num_observations << 32));
      digests.push_back(h);
      digests.push_back(fnv1a(pk.observations, sizeof(double) * 8 * pk.num_observations));
      slslam_summary sum;
      const int rc = solve(pk, max_iter, &sum);
      if (rc) { std::fprintf(stderr, "window solve of keyframe %d: %s\n", k, slslam_status_string(rc)); return rc; }
      if (slslam_unpack_window(&pk, kfs.data(), (int)kfs.size(), lmv.data(), (int)lmv.size())) return 72;
      for (int j = 0; j <= k; ++j) kfT[j] = kfs[j].T;
      for (const slslam_landmark& l : lmv) std::memcpy(lms[l.id].line, l.line, sizeof(l.line));
      s_it += sum.num_successful_steps + sum.num_unsuccessful_steps;
      s_c0 += sum.initial_cost; s_c1 += sum.final_cost;
    }
    slslam_free_packed_window(&pk);
  }

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror(argv[2]); return 73; }
  for (const slslam_pose& T : kfT) { std::fwrite(T.R, sizeof(double), 9, o); std::fwrite(T.t, sizeof(double), 3, o); }
  const double tail[3] = { s_it, s_c0, s_c1 };
  std::fwrite(tail, sizeof(double), 3, o);
  std::fwrite(digests.data(), sizeof(unsigned long long), digests.size(), o);
  std::fclose(o);
  // save_trajectory after metric_embedding(0): poses relative to keyframe 0
  std::vector<slslam_pose> rooted(kfT.size());
  for (size_t j = 0; j < kfT.size(); ++j) slslam_gc_T_21(&kfT[j], &kfT[0], &rooted[j]);
  return slslam_write_trajectory(argv[3], rooted.data(), (int)rooted.size());
}
