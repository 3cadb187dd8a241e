// debugging aid: device routines compiled for the host, traced next to the oracle
#define _GNU_SOURCE 1
#include <math.h>
#include <vector>
#include <cstdio>
#include "ptx_device.h"
#include "ptx_scene_io.h"
static void affine34(const double*m4,double*m){for(int r=0;r<3;r++)for(int q=0;q<4;q++)m[4*r+q]=m4[4*r+q];}
static void tri_normal(const ptx_object &o, double *n) {
    double e12[3], e13[3];
    for (int q = 0; q < 3; q++) { e12[q] = o.v[1][q] + o.v[0][q] * -1; e13[q] = o.v[2][q] + o.v[0][q] * -1; }
    double x = (e12[1] * e13[2]) - (e13[1] * e12[2]); double y = (e13[0] * e12[2]) - (e12[0] * e13[2]); double z = (e12[0] * e13[1]) - (e13[0] * e12[1]);
    double l = x * x; l += (y * y); l += (z * z); l = 1.0 / sqrt(l); n[0] = x * l; n[1] = y * l; n[2] = z * l;
}
struct Holder { std::vector<DevGeom> g; std::vector<DevMat> m; std::vector<DevLight> l; std::vector<DevImage> im; std::vector<uint8_t> tx; DevScene sc; };
extern "C" void* cpudev_load(const ptx_scene_desc* s){
  Holder*h=new Holder(); int n=s->n_objects; h->g.resize(n); h->m.resize(n); int has=0;
  for(int k=0;k<n;k++){const ptx_object&o=s->objects[k]; DevGeom&g=h->g[k]; memset(&g,0,sizeof g); g.type=o.type;
    if(o.type==PTX_TRIANGLE){for(int v=0;v<3;v++)for(int q=0;q<3;q++)g.m[3*v+q]=o.v[v][q]; tri_normal(o,&g.m[9]);} else affine34(o.Tinv,g.m);
    DevMat&m=h->m[k]; memset(&m,0,sizeof m); for(int q=0;q<3;q++)m.col[q]=o.col[q]; m.diffPct=o.diffPct;m.reflPct=o.reflPct;m.tranPct=o.tranPct;m.refl_sig=o.refl_sig;m.r_index=o.r_index;m.surfaceArea=o.surfaceArea;m.is_light=o.is_light;m.tex_img=o.tex_img;m.normal_img=o.normal_img;m.alpha_img=o.alpha_img; if(o.tranPct>0||o.alpha_img>=0)has=1;}
  h->l.resize(s->n_lights>0?s->n_lights:1); for(int l=0;l<s->n_lights;l++){int oi=s->lights[l]; const ptx_object&o=s->objects[oi]; h->l[l].obj=oi;h->l[l].type=o.type;h->l[l].weight=o.LSweight;affine34(o.T,h->l[l].T);}
  h->im.resize(s->n_images>0?s->n_images:1); for(int q=0;q<s->n_images;q++){const ptx_image&im=s->images[q]; h->im[q].sx=im.sx;h->im[q].sy=im.sy;h->im[q].channels=im.channels;h->im[q].divisor=im.divisor;h->im[q].offset=im.data_offset;}
  h->tx.assign(s->texels,s->texels+s->n_texel_bytes);
  DevScene&d=h->sc; memset(&d,0,sizeof d); d.geom=h->g.data();d.mat=h->m.data();d.lights=h->l.data();d.images=h->im.data();d.texels=h->tx.data(); d.n_objects=n;d.n_lights=s->n_lights;d.n_images=s->n_images;d.has_refraction=has;
  const ptx_camera&cam=s->cam; for(int q=0;q<3;q++){d.cam.e[q]=cam.e[q];d.cam.w[q]=cam.w[q];} affine34(cam.C2W,d.cam.C2W); d.cam.f=cam.f;d.cam.wl=cam.wl;d.cam.wt=cam.wt;d.cam.wsize=cam.wsize;
  for(int r=0;r<3;r++) d.cam.fp[r]=(cam.C2W[4*r+0]*0.0)+(cam.C2W[4*r+1]*0.0)+(cam.C2W[4*r+2]*cam.focus_distance)+(cam.C2W[4*r+3]*1.0);
  return h; }
extern "C" void cpudev_trace(void*hp,const ptx_render_params*p,const int*pi,const int*pj,const int*pk,int n,double*out,int verbose){
  Holder*h=(Holder*)hp; DevRender rp; memset(&rp,0,sizeof rp); rp.sx=p->sx;rp.max_depth=p->max_depth;rp.n_ranks=1;rp.band_rows=p->sx;rp.rows_owned=p->sx;rp.aperture=p->aperture;rp.seed=p->seed;rp.du=h->sc.cam.wsize/(p->sx-1);rp.dv=-h->sc.cam.wsize/(p->sx-1);
  for(int q=0;q<n;q++){Counters c={0,0,0}; Path ps; primary_ray(h->sc,rp,pi[q],pj[q],pk[q],h->sc.has_refraction!=0,ps);
    for(;;){ bool end; switch(p->flags&7){case 0:end=bounce<0>(h->sc,rp,ps,c);break;case 7:end=bounce<7>(h->sc,rp,ps,c);break;default:end=bounce<2>(h->sc,rp,ps,c);}
      if(verbose) printf("  dev depth %d os %d p0 %.17g %.17g %.17g d %.17g %.17g %.17g col %.6g %.6g %.6g I %.6g %.6g %.6g ridx %d\n",ps.depth,ps.os,ps.p0.x,ps.p0.y,ps.p0.z,ps.d.x,ps.d.y,ps.d.z,ps.col.x,ps.col.y,ps.col.z,ps.I.x,ps.I.y,ps.I.z,ps.ridx);
      if(end)break;}
    out[3*q]=ps.I.x;out[3*q+1]=ps.I.y;out[3*q+2]=ps.I.z;}
}
