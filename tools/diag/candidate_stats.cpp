// candidate_stats.cpp -- what-if for a table of first-hit candidates (tools only): how many distinct first-hit wall pixels does the
// beam (start cell x fine direction sector) of a ray have?   g++ -O2 tools/diag/candidate_stats.cpp -o /tmp/cand; /tmp/cand /tmp/track.raw /tmp/poses.bin 128 32   (inputs: tools/model_inputs.py)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#include <algorithm>
int W,H,wpr; std::vector<uint32_t> bits;
static inline int wall(int x,int y){ return (bits[(size_t)y*wpr+(x>>5)]>>(x&31))&1u; }
static long first_hit(double px,double py,double dx,double dy){
  int ix=(int)floor(px), iy=(int)floor(py);
  if(ix<0||iy<0||ix>=W||iy>=H) return -1;
  double ivx=dx!=0?fabs(1/dx):1e300, ivy=dy!=0?fabs(1/dy):1e300;
  int sx=dx<0?-1:1, sy=dy<0?-1:1;
  double tx=(dx<0?(px-ix):(ix+1-px))*ivx, ty=(dy<0?(py-iy):(iy+1-py))*ivy;
  for(;;){ if(wall(ix,iy)) return (long)iy*W+ix;
    if(tx<ty){ ix+=sx; tx+=ivx; } else { iy+=sy; ty+=ivy; }
    if(ix<0||iy<0||ix>=W||iy>=H) return -2; }
}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); int32_t hdr[3]; fread(hdr,4,3,f); W=hdr[0];H=hdr[1];wpr=hdr[2]; bits.resize((size_t)H*wpr); fread(bits.data(),4,bits.size(),f); fclose(f);
  f=fopen(argv[2],"rb"); double ph[6]; fread(ph,8,6,f); int n=(int)ph[0]; std::vector<double> pose((size_t)n*4); fread(pose.data(),8,pose.size(),f); fclose(f);
  const int R=1080, NSEC=argc>3?atoi(argv[3]):128; // slope slices per octant
  const int ncar=argc>4?atoi(argv[4]):48;
  std::vector<long> hist(64,0); long rays=0; std::vector<long> hist_by_group(17*64,0);
  for(int c=0;c<ncar;++c){ const double*p=&pose[(size_t)c*4*(n/ncar)];
    double ch=1-2*p[3]*p[3], sh=2*p[2]*p[3]; double lcx=p[0]+ch*-0.0525, lcy=p[1]+sh*-0.0525;
    double u0=(lcx-ph[3])/ph[1], v0=(ph[4]-lcy)/ph[2];
    for(int j=0;j<R;++j){ double phi=((360.0/R)*j-90.0)*(M_PI/180.0); double bx=sin(phi),by=-cos(phi);
      double dxw=ch*bx-sh*by, dyw=sh*bx+ch*by; double du=dxw/ph[1], dv=-dyw/ph[2]; double pu=u0-0.03*du, pv=v0-0.03*dv;
      int cx=(int)floor(pu), cy=(int)floor(pv);
      // fine sector: octant + slope slice
      double adu=fabs(du),adv=fabs(dv); bool ydom=adv>adu; double mj=ydom?adv:adu, mn=ydom?adu:adv; double slope=mn/mj; int sl=std::min(NSEC-1,(int)(slope*NSEC));
      double s_lo=(double)sl/NSEC, s_hi=(double)(sl+1)/NSEC;
      std::set<long> cand;
      for(int a=0;a<12;++a) for(int b=0;b<12;++b) for(int k=0;k<5;++k){
        double ox=cx+(a+0.5)/12.0 *0.9999+ (a==0?1e-6:0), oy=cy+(b+0.5)/12.0;
        if(a==0) ox=cx+1e-9; if(a==11) ox=cx+1-1e-9; if(b==0) oy=cy+1e-9; if(b==11) oy=cy+1-1e-9;
        double s=s_lo+(s_hi-s_lo)*k/4.0; double dmj=1, dmn=s;
        double ddx=ydom?dmn:dmj, ddy=ydom?dmj:dmn; if(du<0) ddx=-ddx; if(dv<0) ddy=-ddy;
        long h=first_hit(ox,oy,ddx,ddy); cand.insert(h); }
      int nc=(int)cand.size(); hist[std::min(nc,63)]++; ++rays; hist_by_group[(j/64)*64+std::min(nc,63)]++;
    }}
  printf("slices per octant %d (sectors %d): candidates per beam (sampled lower bound), %ld rays\n",NSEC,8*NSEC,rays);
  long cum=0; for(int k=1;k<64;++k){ cum+=hist[k]; if(hist[k]) printf("%2d: %6.2f %%  (cum %6.2f %%)\n",k,100.0*hist[k]/rays,100.0*cum/rays); if(100.0*cum/rays>99.5) break; }
  // groups of 64: fraction of groups whose max <= 4, 6, 8
  for(int lim: {4,6,8,12}){ long ok=0,tot=0; for(int c=0;c<1;++c){} // per ray index group aggregated: approximate by per-ray prob
    double pg=1; (void)pg; }
  for(int g=0;g<17;++g){ long tot=0,le4=0,le8=0; for(int k=0;k<64;++k){ tot+=hist_by_group[g*64+k]; if(k<=4) le4+=hist_by_group[g*64+k]; if(k<=8) le8+=hist_by_group[g*64+k]; } printf("ray group %2d: <=4: %5.1f %%  <=8: %5.1f %%\n",g,100.0*le4/tot,100.0*le8/tot); }
}
