// tools/pack_phases.cpp - where a pack goes under the refill's memory pattern: N distinct copies of a window packed by T threads into one image,
// the packer's phase clock (-DSLS_PACK_TIMING: racy sums over the threads, good to a few per cent) split by phase.
//   g++ -O3 -std=c++17 -pthread -DSLS_PACK_TIMING -I slslam_amd/csrc tools/pack_phases.cpp slslam_amd/csrc/lba_pack.cpp -o pack_phases ; ./pack_phases <threads> <copies>   (window dumped by tools/pack_bench.sh)
#include "lba_pack.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>
using namespace slslam;
namespace slslam { extern double g_pack_phase_ms[8]; }
template<class T> std::vector<T> rd(const char*f){std::ifstream s(f,std::ios::binary);s.seekg(0,std::ios::end);size_t n=s.tellg();s.seekg(0);std::vector<T> v(n/sizeof(T));s.read((char*)v.data(),n);return v;}
int main(int argc,char**argv){
  const int T=argc>1?atoi(argv[1]):1, N=argc>2?atoi(argv[2]):512;
  auto h=rd<int>("hdr.bin");auto cam=rd<int>("cam.bin");auto line=rd<int>("line.bin");auto fx=rd<int>("fixed.bin");auto ob=rd<double>("obs.bin");auto par=rd<double>("par.bin");
  const size_t M=(size_t)h[2];
  struct Copy { std::vector<int> cam,line,fx; std::vector<double> ob,par; slslam_lba_window w; };
  std::vector<Copy> cs((size_t)N);
  for(auto&c:cs){ c.cam=cam;c.line=line;c.fx=fx;c.ob=ob;c.par=par; c.w.num_cameras=h[0];c.w.num_lines=h[1];c.w.num_observations=h[2];c.w.camera_index=c.cam.data();c.w.line_index=c.line.data();c.w.fixed_index=c.fx.data();c.w.observations=c.ob.data();c.w.parameters=c.par.data(); }
  const size_t stride=M*(size_t)N;
  double* dest=(double*)malloc(8*stride*sizeof(double)); memset(dest,0,8*stride*sizeof(double));
  std::vector<PackedWindow> out((size_t)N);
  for(int r=0;r<3;++r){
    for(int q=0;q<8;++q) g_pack_phase_ms[q]=0;
    std::atomic<int> next{0};
    auto t0=std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for(int t=0;t<T;++t) th.emplace_back([&]{ for(;;){ int i=next.fetch_add(1); if(i>=N)break; ObPlanes d; for(int q=0;q<4;++q) d.plane[q]=dest+((size_t)q*stride+(size_t)i*M)*2; pack_window(&cs[(size_t)i].w,&out[(size_t)i],1,&d);} });
    for(auto&x:th)x.join();
    const double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
    printf("threads %2d, %d windows: %.2f ms wall = %.3f thread-ms per window;", T, N, ms, ms*T/N);
    const char* nm[6]={"validate+counts","rows","line order","obs sort","obs gather","tiles"};
    for(int q=0;q<6;++q) printf("  %s %.3f", nm[q], g_pack_phase_ms[q]/N);
    printf("\n");
  }
}
