/*
 * ptx_kit.h -- host-side scene kit: the types and functions a buildScene()
 * file is written against.
 *
 * Names, argument lists and meaning follow the reference's scene API
 * (PathTracer.h:42-182 for the types, utils_path.h:26-300 and PathTracer.c:42
 * for the functions), so a scene file written for the reference compiles
 * against this header unchanged.  The implementation behind it is this
 * project's own; the hot path never sees these structs -- ptxh_flatten()
 * turns them into the flat description of include/ptx_scene.h.
 */
#ifndef PTX_KIT_H
#define PTX_KIT_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptx_scene.h"

#define PI 3.14159265354 /* reference PathTracer.h:27 */
#define TOL 0.000000001  /* reference PathTracer.h:28 */

struct image {
    void *rgbdata; /* double[sx*sy*layers] in [0,1] for textures */
    int sx, sy;
};

struct textureNode {
    char name[1024];
    int type; /* 1 texture, 2 normal map, 3 alpha map */
    struct image *im;
    unsigned char *bytes; /* the file's samples as read (what the GPU is given) */
    double divisor;       /* bytes -> rgbdata divisor */
    struct textureNode *next;
};

struct point3D {
    double px, py, pz, pw;
};

struct colourRGB {
    double R, G, B;
};

struct Box {
    double xl, xu, yl, yu, zl, zu;
};

enum ptx_prim_kind { KIT_PLANE = 0, KIT_SPHERE, KIT_CYL, KIT_TRIANGLE, KIT_CUBE, KIT_ROOM };

struct object3D {
    struct colourRGB col;
    struct point3D v1, v2, v3; /* triangle vertices, world space */
    double T[4][4];
    double Tinv[4][4];
    int kind; /* enum ptx_prim_kind (the reference stores an intersect function pointer instead) */
    struct image *texImg, *normalImg, *alphaImg;
    struct Box box;
    double diffPct, reflPct, tranPct;
    double refl_sig, r_index;
    int isLightSource;
    double surfaceArea;
    double LSweight;
    int frontAndBack;
    int isCSG;
    int insert_id; /* order of the insertObject() call */
    struct object3D *LSnext;
    struct object3D *next;
};

struct view {
    struct point3D e, u, v, w;
    double f, focus_distance;
    double wl, wt, wsize;
    double W2C[4][4], C2W[4][4];
};

/* The scene lives in these globals, as in the reference (PathTracer.c:34-38). */
extern struct object3D *object_list;
extern struct object3D *light_list;
extern struct textureNode *texture_list;
extern struct view *cam;

void buildScene(void); /* supplied by the scene file */

/* ---- small vector / matrix helpers scene files may use (utils_path.h:26-230) */
static inline void matMult(double A[4][4], double B[4][4]) { /* B = A*B */
    double C[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            C[i][j] = (A[i][0] * B[0][j]) + (A[i][1] * B[1][j]) + (A[i][2] * B[2][j]) + (A[i][3] * B[3][j]);
    memcpy(B, C, sizeof(C));
}
static inline void divideHCoords(struct point3D *v) {
    v->pw = fabs(v->pw);
    if (v->pw != 1) {
        v->px /= v->pw;
        v->py /= v->pw;
        v->pz /= v->pw;
        v->pw = 1;
    }
}
static inline void matVecMult(double A[4][4], struct point3D *pt) {
    struct point3D r;
    r.px = (A[0][0] * pt->px) + (A[0][1] * pt->py) + (A[0][2] * pt->pz) + (A[0][3] * pt->pw);
    r.py = (A[1][0] * pt->px) + (A[1][1] * pt->py) + (A[1][2] * pt->pz) + (A[1][3] * pt->pw);
    r.pz = (A[2][0] * pt->px) + (A[2][1] * pt->py) + (A[2][2] * pt->pz) + (A[2][3] * pt->pw);
    r.pw = (A[3][0] * pt->px) + (A[3][1] * pt->py) + (A[3][2] * pt->pz) + (A[3][3] * pt->pw);
    *pt = r;
    divideHCoords(pt);
}
static inline void normalize(struct point3D *v) {
    divideHCoords(v);
    double l = v->px * v->px;
    l += v->py * v->py;
    l += v->pz * v->pz;
    l = 1.0 / sqrt(l);
    v->px *= l;
    v->py *= l;
    v->pz *= l;
}
static inline double dot(struct point3D *u, struct point3D *v) {
    divideHCoords(u);
    divideHCoords(v);
    return (u->px * v->px) + (u->py * v->py) + (u->pz * v->pz);
}
static inline struct point3D *cross(struct point3D *u, struct point3D *v) { /* caller frees */
    struct point3D *c = (struct point3D *)calloc(1, sizeof(*c));
    divideHCoords(u);
    divideHCoords(v);
    c->px = (u->py * v->pz) - (v->py * u->pz);
    c->py = (v->px * u->pz) - (u->px * v->pz);
    c->pz = (u->px * v->py) - (v->px * u->py);
    c->pw = 1;
    return c;
}

/* ---- transforms (utils_path.h:86-99) */
void invert(double *T, double *Tinv);
void RotateXMat(double T[4][4], double theta);
void RotateYMat(double T[4][4], double theta);
void RotateZMat(double T[4][4], double theta);
void TranslateMat(double T[4][4], double tx, double ty, double tz);
void ScaleMat(double T[4][4], double sx, double sy, double sz);
void RotateX(struct object3D *o, double theta);
void RotateY(struct object3D *o, double theta);
void RotateZ(struct object3D *o, double theta);
void Translate(struct object3D *o, double tx, double ty, double tz);
void Scale(struct object3D *o, double sx, double sy, double sz);
void printmatrix(double mat[4][4]);

/* ---- object constructors (utils_path.h:239-244) */
struct object3D *newPlane(double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);
struct object3D *newSphere(double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);
struct object3D *newCyl(double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);
struct object3D *newTriangle(struct point3D v1, struct point3D v2, struct point3D v3, double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);
struct object3D *newCube(double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);
struct object3D *newRoom(double diffPct, double reflPct, double tranPct, double r, double g, double b, double refl_sig, double r_index);

/* ---- lists, textures, meshes, camera (utils_path.h:272-300, PathTracer.c:42) */
void insertObject(struct object3D *o, struct object3D **list);
void loadTexture(struct object3D *o, const char *filename, int type, struct textureNode **t_list);
void loadObj(const char *fname, double diff, double refl, double tran, double r, double g, double b, double refl_sig, double r_index, struct object3D **object_list, double T[4][4]);
struct view *setupView(struct point3D *e, struct point3D *g, struct point3D *up, double f, double wl, double wt, double wsize, double focus_distance);
struct image *readPPMimage(const char *filename);
struct image *readPGMimage(const char *filename);
struct image *newImage(int size_x, int size_y);
void deleteImage(struct image *im);
void dataOutput(double *im, int sx, char *name);
void ptxh_write_output(const double *im, const unsigned char *bits, int sx, const char *name, double mi, double mx,
                       double low_cut, double high_cut);
void cleanup(struct object3D *o_list, struct textureNode *t_list);

/* ---- what sits between buildScene() and the GPU library */

/* Forgets the current scene (frees objects, textures, camera). */
void ptxh_reset(void);

/* Everything the reference's main() does between buildScene() and the sample loop that changes what can be hit and
 * how lights are weighted (PathTracer.c:566, :600-611): builds the BoxTree over object_list, normalises the light
 * weights, and writes the flat description.  The arrays it points to belong to the kit and stay valid until the
 * next ptxh_flatten() or ptxh_reset().  Returns 0, or -1 if there is no camera or no object. */
int ptxh_flatten(ptx_scene_desc *out);

/* Number of objects insertObject() has seen for the current scene. */
int ptxh_inserted_count(void);

#endif
