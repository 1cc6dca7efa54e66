// tools/pack_bench.cpp - how the host-side build stage (pack_window: the LBAProblem::build stage, reference src/lba_problem.cpp:54-93) scales
// over host threads on THIS box: the same 2000-line window packed `reps` times by T threads, straight into observation planes as a
// refill does.  tools/pack_bench.sh dumps the window, compiles and runs it for T = 1, 2, 4, 8, 16, 32.
#include "lba_pack.h"
#include <cstdio>
#include <chrono>
#include <fstream>
#include <thread>
using namespace slslam;
template<class T> std::vector<T> rd(const char*f){std::ifstream s(f,std::ios::binary);s.seekg(0,std::ios::end);size_t n=s.tellg();s.seekg(0);std::vector<T> v(n/sizeof(T));s.read((char*)v.data(),n);return v;}
int main(int argc,char**argv){
  int T=argc>1?atoi(argv[1]):1;
  auto h=rd<int>("hdr.bin");auto cam=rd<int>("cam.bin");auto line=rd<int>("line.bin");auto fx=rd<int>("fixed.bin");auto ob=rd<double>("obs.bin");auto par=rd<double>("par.bin");
  slslam_lba_window w{}; w.num_cameras=h[0];w.num_lines=h[1];w.num_observations=h[2];w.camera_index=cam.data();w.line_index=line.data();w.fixed_index=fx.data();w.observations=ob.data();w.parameters=par.data();
  const int reps=200;
  auto t0=std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for(int t=0;t<T;++t) th.emplace_back([&]{ PackedWindow Q; std::vector<double> dest(8*(size_t)w.num_observations); ObPlanes d; for(int q=0;q<4;++q) d.plane[q]=dest.data()+2*(size_t)q*w.num_observations; for(int r=0;r<reps;++r) pack_window(&w,&Q,1,&d);});
  for(auto&x:th)x.join();
  double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
  printf("threads %d: %.3f ms per pack per thread, aggregate %.3f ms per pack\n",T,ms/reps,ms/reps/T);
}
