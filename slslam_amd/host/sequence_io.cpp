// slslam_amd/host/sequence_io.cpp — see sequence_io.h.
#include "sequence_io.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace {

struct Obs8 { double v[8]; };

// one text line "id x0 .. x7 [rest]" -> false on a blank line (the reference would dereference NULL there)
bool parse_line(char* line, int* id, Obs8* o) {
  char* tok = std::strtok(line, " ");
  if (!tok || *tok == '\n' || *tok == '\r' || *tok == 0) return false;
  *id = std::atoi(tok);
  for (int q = 0; q < 8; ++q) {
    tok = std::strtok(nullptr, " ");
    o->v[q] = tok ? std::atof(tok) : 0.0;
  }
  return true;
}

int finish_frame(const std::map<int, Obs8>& cur, slslam_frame* out) {
  const int n = (int)cur.size();
  out->num_lines = n;
  out->ids = (int*)std::malloc(sizeof(int) * (n ? n : 1));
  out->observations = (double*)std::malloc(sizeof(double) * 8 * (n ? n : 1));
  if (!out->ids || !out->observations) return 2;
  int i = 0;
  for (const auto& kv : cur) {
    out->ids[i] = kv.first;
    for (int q = 0; q < 8; ++q) out->observations[8 * i + q] = kv.second.v[q];
    ++i;
  }
  return 0;
}

void insert_obs(std::map<int, Obs8>& cur, int id, Obs8 o, const slslam_intrinsics* K, const std::map<int, int>& alias) {
  // insert_curr_obs (slam.cpp:121-135): obs = obs / f - c / f, x with (fx1, cx1), y with (fy1, cy1), for both cameras
  for (int q = 0; q < 8; ++q) {
    const bool is_x = (q % 2) == 0;
    o.v[q] = is_x ? o.v[q] / K->fx - K->cx / K->fx : o.v[q] / K->fy - K->cy / K->fy;
  }
  auto it = alias.find(id);
  if (it != alias.end()) id = it->second;
  cur.insert(std::make_pair(id, o));
}

std::string fmt_g(double v) {     // operator<<(ostream&, double) with default flags = %g, precision 6
  char b[64];
  std::snprintf(b, sizeof b, "%g", v);
  return b;
}

}  // namespace

extern "C" {

int slslam_parse_frame_text(const char* text, size_t len, const slslam_intrinsics* K, const int* alias_from,
                            const int* alias_to, int n_alias, slslam_frame* out) {
  if (!text || !K || !out) return 2;
  std::map<int, int> alias;
  for (int i = 0; i < n_alias; ++i) alias[alias_from[i]] = alias_to[i];
  std::map<int, Obs8> cur;
  size_t p = 0;
  char line[256];
  while (p < len) {
    size_t e = p;
    while (e < len && text[e] != '\n') ++e;
    if (e - p > 255) break;                  // getline(line, 256) stores up to 255 characters; a longer line sets failbit: the read ends
    std::memcpy(line, text + p, e - p);
    line[e - p] = 0;
    int id;
    Obs8 o;
    if (parse_line(line, &id, &o)) insert_obs(cur, id, o, K, alias);
    p = e + 1;
  }
  return finish_frame(cur, out);
}

int slslam_read_frame_file(const char* path, const slslam_intrinsics* K, const int* alias_from, const int* alias_to,
                           int n_alias, slslam_frame* out) {
  if (!path || !K || !out) return 2;
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return 1;
  std::string text;
  char buf[4096];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
  std::fclose(f);
  return slslam_parse_frame_text(text.data(), text.size(), K, alias_from, alias_to, n_alias, out);
}

int slslam_frame_path(const char* obs_dir, int frame_id, char* buf, size_t cap) {
  const int n = std::snprintf(buf, cap, "%s/%04d.txt", obs_dir, frame_id);
  return n > 0 && (size_t)n < cap ? 0 : 2;
}

void slslam_free_frame(slslam_frame* f) {
  if (!f) return;
  std::free(f->ids); std::free(f->observations);
  f->ids = nullptr; f->observations = nullptr; f->num_lines = 0;
}

int slslam_metric_embedding(int root, int n, const int* kf_ids, const int* nbr_ptr, const int* nbr,
                            const slslam_me_edge* edges, int num_edges, slslam_pose* T, int* order_ids,
                            double* order_dist, int* n_embedded) {
  std::map<int, int> index;
  for (int i = 0; i < n; ++i) index[kf_ids[i]] = i;
  if (!index.count(root)) return 1;
  std::map<std::pair<int, int>, const slslam_pose*> emap;
  for (int e = 0; e < num_edges; ++e) emap[std::make_pair(edges[e].from, edges[e].to)] = &edges[e].T;
  const slslam_pose ident = { { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, { 0, 0, 0 } };
  std::multimap<double, int> m;
  m.insert(std::make_pair(0.0, root));                       // :1321-1323
  T[index[root]] = ident;
  std::set<int> embedded;
  embedded.insert(root);
  int prev = -1, cnt = 0;
  while (!m.empty()) {                                       // :1331-1361
    auto mit = m.begin();
    const double d = mit->first;
    const int start = mit->second;
    const int si = index[start];
    const slslam_pose Ts = T[si];
    if (order_ids) order_ids[cnt] = start;
    if (order_dist) order_dist[cnt] = d;
    ++cnt;
    m.erase(mit);
    for (int q = nbr_ptr[si]; q < nbr_ptr[si + 1]; ++q) {
      const int end = nbr[q];
      if (end == prev) continue;
      if (embedded.count(end)) continue;
      auto ix = index.find(end);
      if (ix == index.end()) continue;                       // the reference would create the keyframe entry; not representable here
      auto eit = emap.find(std::make_pair(start, end));
      const slslam_pose* Te = eit != emap.end() ? eit->second : &ident;
      const double nd = std::sqrt(Te->t[0] * Te->t[0] + Te->t[1] * Te->t[1] + Te->t[2] * Te->t[2]);
      m.insert(std::make_pair(d + nd, end));
      slslam_gc_T_20(Te, &Ts, &T[ix->second]);
      embedded.insert(end);
    }
    prev = start;
  }
  if (n_embedded) *n_embedded = cnt;
  return 0;
}

int slslam_format_trajectory_line(int index, const slslam_pose* T_kf, char* buf, size_t cap) {
  slslam_pose Ti;
  double w[3];
  slslam_gc_T_inv(T_kf, &Ti);                                // :1476-1477
  slslam_gc_R_to_rodrigues(Ti.R, w);                         // :1489
  const std::string s = std::to_string(index) + "\t" + fmt_g(Ti.t[2]) + "\t" + fmt_g(-Ti.t[0]) + "\t" + fmt_g(-Ti.t[1]) + "\t" +
                        fmt_g(w[0]) + "\t" + fmt_g(w[1]) + "\t" + fmt_g(w[2]) + "\n";   // :1490-1491
  if (s.size() + 1 > cap) return 2;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

int slslam_write_trajectory(const char* path, const slslam_pose* T, int n) {
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  char line[512];
  for (int i = 0; i < n; ++i) {
    if (slslam_format_trajectory_line(i, &T[i], line, sizeof line)) { std::fclose(f); return 2; }
    std::fputs(line, f);
  }
  std::fclose(f);
  return 0;
}

int slslam_landmark_endpoints(const double line[6], const double tt[2], const slslam_pose* T_init, double ep[6]) {
  const double* p = line; const double* v = line + 3;        // :1441-1447
  const double nx = p[1] * v[2] - p[2] * v[1], ny = p[2] * v[0] - p[0] * v[2], nz = p[0] * v[1] - p[1] * v[0];
  const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double p0[3] = { (v[1] * nz - v[2] * ny) / vv, (v[2] * nx - v[0] * nz) / vv, (v[0] * ny - v[1] * nx) / vv };
  const double vn = std::sqrt(vv);
  slslam_pose Ti;
  slslam_gc_T_inv(T_init, &Ti);                              // gc_poit_from_pose = gc_point_to_pose(gc_T_inv(T), .)
  for (int e = 0; e < 2; ++e) {
    const double pc[3] = { p0[0] + v[0] / vn * tt[e], p0[1] + v[1] / vn * tt[e], p0[2] + v[2] / vn * tt[e] };
    for (int r = 0; r < 3; ++r)
      ep[3 * e + r] = Ti.R[3 * r] * pc[0] + Ti.R[3 * r + 1] * pc[1] + Ti.R[3 * r + 2] * pc[2] + Ti.t[r];
  }
  return 0;
}

int slslam_write_landmarks(const char* path, const double* lines, const double* tt, const slslam_pose* T_init, int n) {
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return 1;
  for (int i = 0; i < n; ++i) {
    double e[6];
    slslam_landmark_endpoints(lines + 6 * (size_t)i, tt + 2 * (size_t)i, &T_init[i], e);
    const std::string s = fmt_g(e[2]) + "\t" + fmt_g(-e[1]) + "\t" + fmt_g(e[0]) + "\t" + fmt_g(e[5]) + "\t" + fmt_g(-e[4]) +
                          "\t" + fmt_g(e[3]) + "\n";         // :1463-1464
    std::fputs(s.c_str(), f);
  }
  std::fclose(f);
  return 0;
}

}  // extern "C"
