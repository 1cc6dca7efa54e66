// tools/pack_bench2.cpp - the packer under the memory pattern of a refill (tools/pack_bench.cpp packs ONE cache-hot window into thread-local
// planes): N distinct copies of the window's arrays made by the main thread, packed round-robin by T threads into ONE destination
// image made by the main thread - pageable or pinned (hipHostMalloc), as slslam_lba_batch_refill does.  Separates what the packer
// costs from what the memory system of the box charges for it.
//   hipcc -O3 -std=c++17 -I slslam_amd/csrc tools/pack_bench2.cpp slslam_amd/csrc/lba_pack.cpp -o pack_bench2 ; ./pack_bench2 <threads> <pinned 0|1> <copies>
#include <hip/hip_runtime.h>
#include "lba_pack.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>
using namespace slslam;
template<class T> std::vector<T> rd(const char*f){std::ifstream s(f,std::ios::binary);s.seekg(0,std::ios::end);size_t n=s.tellg();s.seekg(0);std::vector<T> v(n/sizeof(T));s.read((char*)v.data(),n);return v;}
int main(int argc,char**argv){
  const int T=argc>1?atoi(argv[1]):1, pinned=argc>2?atoi(argv[2]):0, N=argc>3?atoi(argv[3]):256, reps=3;
  auto h=rd<int>("hdr.bin");auto cam=rd<int>("cam.bin");auto line=rd<int>("line.bin");auto fx=rd<int>("fixed.bin");auto ob=rd<double>("obs.bin");auto par=rd<double>("par.bin");
  const size_t M=(size_t)h[2];
  struct Copy { std::vector<int> cam,line,fx; std::vector<double> ob,par; slslam_lba_window w; };
  std::vector<Copy> cs((size_t)N);
  for(auto&c:cs){ c.cam=cam;c.line=line;c.fx=fx;c.ob=ob;c.par=par; c.w.num_cameras=h[0];c.w.num_lines=h[1];c.w.num_observations=h[2];c.w.camera_index=c.cam.data();c.w.line_index=c.line.data();c.w.fixed_index=c.fx.data();c.w.observations=c.ob.data();c.w.parameters=c.par.data(); }
  const size_t stride=M*(size_t)N;
  double* dest=nullptr;
  if(pinned){ if(hipHostMalloc((void**)&dest, 8*stride*sizeof(double), hipHostMallocDefault)!=hipSuccess){printf("hipHostMalloc failed\n");return 1;} }
  else dest=(double*)malloc(8*stride*sizeof(double));
  memset(dest,0,8*stride*sizeof(double));
  std::vector<PackedWindow> out((size_t)N);
  double best=1e30;
  for(int r=0;r<reps;++r){
    std::atomic<int> next{0};
    auto t0=std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for(int t=0;t<T;++t) th.emplace_back([&]{ for(;;){ int i=next.fetch_add(1); if(i>=N)break; ObPlanes d; for(int q=0;q<4;++q) d.plane[q]=dest+((size_t)q*stride+(size_t)i*M)*2; pack_window(&cs[(size_t)i].w,&out[(size_t)i],1,&d);} });
    for(auto&x:th)x.join();
    best=std::min(best,std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count());
  }
  printf("threads %2d  %s dest  %d distinct windows: %.2f ms per %d = %.4f ms per window aggregate (%.3f thread-ms per window)\n",T,pinned?"pinned  ":"pageable",N,best,N,best/N,best*T/N);
}
