// tools/pack_stats.cpp - tile statistics of a packed window (host only): pair passes per tile, how many lane slots a
// "two items of one camera pair per lane" (rank-8) scheme would need, run lengths.  Driven by tools/pack_stats.py, which dumps
// a synthetic window to the working directory, compiles this file against slslam_amd/csrc/lba_pack.cpp and runs it.
#include "lba_pack.h"
#include <cstdio>
#include <map>
#include <fstream>
using namespace slslam;
template<class T> std::vector<T> rd(const char*f){std::ifstream s(f,std::ios::binary);s.seekg(0,std::ios::end);size_t n=s.tellg();s.seekg(0);std::vector<T> v(n/sizeof(T));s.read((char*)v.data(),n);return v;}
int main(){
  auto h=rd<int>("hdr.bin");auto cam=rd<int>("cam.bin");auto line=rd<int>("line.bin");auto fx=rd<int>("fixed.bin");auto ob=rd<double>("obs.bin");auto par=rd<double>("par.bin");
  slslam_lba_window w{}; w.num_cameras=h[0];w.num_lines=h[1];w.num_observations=h[2];w.camera_index=cam.data();w.line_index=line.data();w.fixed_index=fx.data();w.observations=ob.data();w.parameters=par.data();
  PackedWindow P; int st=pack_window(&w,&P); printf("status %d tiles %zu items %zu\n",st,P.tiles.size(),P.items.size()/2);
  long passes=0,p8=0,nit=0; int hist[8]={0}; long run8=0, multi=0, maxrun_hist[17]={0};
  long slots_tot=0;
  for(size_t t=0;t<P.tiles.size();++t){const Tile&T=P.tiles[t]; nit+=T.nitems; passes+=(T.nitems+63)/64; hist[std::min(7,(T.nitems+63)/64)]++;
    if(tile_max_run(T.flags)>8)run8++; if(T.flags&1)multi++; maxrun_hist[tile_max_run(T.flags)]++;
    std::map<int,int> cnt; const uint16_t*lm=&P.lane_map[64*t];
    // camera of lane
    int lanecam[64]; for(int l=0;l<64;++l){int m=lm[l]; lanecam[l]=-1; if((m&0xff)!=0xff){int s=T.line_begin+(m&0xff); int j=m>>8; int k=P.line_ptr[s+1]-P.line_ptr[s]; if(j<k) lanecam[l]=P.cam_cf[P.ob_cam[P.line_ptr[s]+j]];}}
    for(int i=0;i<T.nitems;++i){int li=P.items[2*(T.item_off+i)],lj=P.items[2*(T.item_off+i)+1]; cnt[lanecam[li]*32+lanecam[lj]]++;}
    int slots=0; for(auto&kv:cnt) slots+=(kv.second+1)/2; slots_tot+=slots; p8+=(slots+63)/64;
  }
  printf("items/tile %.1f passes/tile %.3f  rank8 slots/tile %.1f passes/tile %.3f  P(maxrun>8) %.3f P(multirow) %.3f\n",(double)nit/P.tiles.size(),(double)passes/P.tiles.size(),(double)slots_tot/P.tiles.size(),(double)p8/P.tiles.size(),(double)run8/P.tiles.size(),(double)multi/P.tiles.size());
  for(int i=0;i<8;++i)printf("passes=%d: %d tiles\n",i,hist[i]);
  for(int i=0;i<17;++i)printf("maxrun=%d: %ld\n",i,maxrun_hist[i]);
}
