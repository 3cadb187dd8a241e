/*
 * ptx_kit.c -- host-side scene kit: object constructors, transforms, lists,
 * texture and mesh loading, camera.  Runs once per scene, on the CPU, before
 * anything is uploaded; behaviour follows the reference routines cited at
 * each function (what a scene file observes is the same), the code is this
 * project's own.
 */
#include "ptx_kit.h"

struct object3D *object_list = NULL;
struct object3D *light_list = NULL;
struct textureNode *texture_list = NULL;
struct view *cam = NULL;

static int g_inserted = 0;

int ptxh_inserted_count(void) { return g_inserted; }
void ptxh_reset_counter(void) { g_inserted = 0; }

/* ------------------------------------------------------------ transforms */

static void set_identity(double M[4][4]) {
    memset(M, 0, 16 * sizeof(double));
    M[0][0] = M[1][1] = M[2][2] = M[3][3] = 1.0;
}

/* Smallest singular value of a 4x4 matrix: square root of the smallest eigenvalue of M^T M, by cyclic Jacobi. */
static double smallest_singular_value(const double *M) {
    double A[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += M[4 * k + i] * M[4 * k + j];
            A[i][j] = s;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) {
                if (A[p][q] == 0) continue;
                double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
            }
    }
    double mn = A[0][0];
    for (int i = 1; i < 4; i++)
        if (A[i][i] < mn) mn = A[i][i];
    return mn > 0 ? sqrt(mn) : 0.0;
}

/* Inverse of a 4x4 matrix.  Like the reference's invert() (utils_path.c:1434-1467, which goes through an SVD) it
 * returns the identity when a singular value is below 1e-9; otherwise Gauss-Jordan elimination with full row
 * pivoting in extended precision. */
void invert(double *T, double *Tinv) {
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (!(smallest_singular_value(T) >= 1e-9)) {
        memcpy(Tinv, eye, sizeof(eye));
        return;
    }
    long double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = T[4 * i + j];
            a[i][4 + j] = i == j ? 1.0L : 0.0L;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (fabsl(a[r][col]) > fabsl(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0L) {
            memcpy(Tinv, eye, sizeof(eye));
            return;
        }
        if (piv != col)
            for (int j = 0; j < 8; j++) {
                long double t = a[col][j];
                a[col][j] = a[piv][j];
                a[piv][j] = t;
            }
        long double inv = 1.0L / a[col][col];
        for (int j = 0; j < 8; j++) a[col][j] *= inv;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            long double f = a[r][col];
            if (f == 0.0L) continue;
            for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Tinv[4 * i + j] = (double)a[i][4 + j];
}

/* axis 0,1,2 = X,Y,Z: the rotation matrices of utils_path.c:1469-1575 */
static void rotation(double R[4][4], int axis, double theta) {
    double c = cos(theta), s = sin(theta);
    set_identity(R);
    int a = (axis + 1) % 3, b = (axis + 2) % 3;
    R[a][a] = c;
    R[a][b] = -s;
    R[b][a] = s;
    R[b][b] = c;
}
void RotateXMat(double T[4][4], double theta) { double R[4][4]; rotation(R, 0, theta); matMult(R, T); }
void RotateYMat(double T[4][4], double theta) { double R[4][4]; rotation(R, 1, theta); matMult(R, T); }
void RotateZMat(double T[4][4], double theta) { double R[4][4]; rotation(R, 2, theta); matMult(R, T); }
void TranslateMat(double T[4][4], double tx, double ty, double tz) {
    double M[4][4];
    set_identity(M);
    M[0][3] = tx;
    M[1][3] = ty;
    M[2][3] = tz;
    matMult(M, T);
}
void ScaleMat(double T[4][4], double sx, double sy, double sz) {
    double M[4][4];
    set_identity(M);
    M[0][0] = sx;
    M[1][1] = sy;
    M[2][2] = sz;
    matMult(M, T);
}
void RotateX(struct object3D *o, double theta) { RotateXMat(o->T, theta); }
void RotateY(struct object3D *o, double theta) { RotateYMat(o->T, theta); }
void RotateZ(struct object3D *o, double theta) { RotateZMat(o->T, theta); }
void Translate(struct object3D *o, double tx, double ty, double tz) { TranslateMat(o->T, tx, ty, tz); }
void Scale(struct object3D *o, double sx, double sy, double sz) {
    ScaleMat(o->T, sx, sy, sz);
    o->LSweight *= (sx * sy * sz); /* the light weight follows the object's size, utils_path.c:1641 */
}
void printmatrix(double mat[4][4]) {
    fprintf(stderr, "Matrix contains:\n");
    for (int i = 0; i < 4; i++) fprintf(stderr, "%f %f %f %f\n", mat[i][0], mat[i][1], mat[i][2], mat[i][3]);
}

/* ---------------------------------------------------------- constructors */

/* Defaults per primitive kind (utils_path.c:151-433): canonical bounding box and initial light weight (the
 * canonical plane has area 4; a cylinder's weight is never set there and stays 0). */
static struct object3D *new_object(int kind, const char *what, double diffPct, double reflPct, double tranPct, double r,
                                   double g, double b, double refl_sig, double r_index) {
    struct object3D *o = (struct object3D *)calloc(1, sizeof(*o));
    if (!o) {
        fprintf(stderr, "Unable to allocate new %s, out of memory!\n", what);
        return NULL;
    }
    o->kind = kind;
    o->diffPct = diffPct;
    o->reflPct = reflPct;
    o->tranPct = tranPct;
    o->col.R = r;
    o->col.G = g;
    o->col.B = b;
    o->refl_sig = refl_sig;
    o->r_index = r_index;
    set_identity(o->T);
    set_identity(o->Tinv);
    o->frontAndBack = kind == KIT_PLANE;
    o->LSweight = kind == KIT_PLANE ? 4.0 : (kind == KIT_CYL ? 0.0 : 1.0);
    o->box.xl = o->box.yl = -1;
    o->box.xu = o->box.yu = 1;
    o->box.zl = kind == KIT_PLANE ? 0 : -1;
    o->box.zu = kind == KIT_PLANE ? 0 : 1;
    o->insert_id = -1;
    return o;
}
struct object3D *newPlane(double d, double rf, double t, double r, double g, double b, double sig, double ri) {
    return new_object(KIT_PLANE, "plane", d, rf, t, r, g, b, sig, ri);
}
struct object3D *newSphere(double d, double rf, double t, double r, double g, double b, double sig, double ri) {
    return new_object(KIT_SPHERE, "sphere", d, rf, t, r, g, b, sig, ri);
}
struct object3D *newCyl(double d, double rf, double t, double r, double g, double b, double sig, double ri) {
    return new_object(KIT_CYL, "cylinder", d, rf, t, r, g, b, sig, ri);
}
struct object3D *newCube(double d, double rf, double t, double r, double g, double b, double sig, double ri) {
    return new_object(KIT_CUBE, "cube", d, rf, t, r, g, b, sig, ri);
}
struct object3D *newRoom(double d, double rf, double t, double r, double g, double b, double sig, double ri) {
    return new_object(KIT_ROOM, "room", d, rf, t, r, g, b, sig, ri);
}
static double min3(double a, double b, double c) { double m = a > b ? b : a; return m > c ? c : m; }
static double max3(double a, double b, double c) { double m = a < b ? b : a; return m < c ? c : m; }
struct object3D *newTriangle(struct point3D v1, struct point3D v2, struct point3D v3, double d, double rf, double t,
                             double r, double g, double b, double sig, double ri) {
    struct object3D *o = new_object(KIT_TRIANGLE, "tri", d, rf, t, r, g, b, sig, ri);
    if (!o) return NULL;
    o->v1 = v1;
    o->v2 = v2;
    o->v3 = v3;
    /* the box comes straight from the (world-space) vertices, utils_path.c:334-340 */
    o->box.xl = min3(v1.px, v2.px, v3.px);
    o->box.xu = max3(v1.px, v2.px, v3.px);
    o->box.yl = min3(v1.py, v2.py, v3.py);
    o->box.yu = max3(v1.py, v2.py, v3.py);
    o->box.zl = min3(v1.pz, v2.pz, v3.pz);
    o->box.zu = max3(v1.pz, v2.pz, v3.pz);
    return o;
}

/* Axis-aligned box of the eight transformed corners (boxtransform, magic.c:397-422). */
static struct Box transformed_box(struct Box b, double T[4][4]) {
    struct Box r = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int c = 0; c < 8; c++) {
        struct point3D p = {c & 4 ? b.xu : b.xl, c & 2 ? b.yu : b.yl, c & 1 ? b.zu : b.zl, 1};
        matVecMult(T, &p);
        if (p.px < r.xl) r.xl = p.px;
        if (p.px > r.xu) r.xu = p.px;
        if (p.py < r.yl) r.yl = p.py;
        if (p.py > r.yu) r.yu = p.py;
        if (p.pz < r.zl) r.zl = p.pz;
        if (p.pz > r.zu) r.zu = p.pz;
    }
    return r;
}

/* insertObject (utils_path.c:125-149): the first object stays at the head of the list, every later one goes right
 * behind the head; lights are threaded through LSnext the same way.  The transform must be final here: the world
 * box of every non-triangle is computed now. */
void insertObject(struct object3D *o, struct object3D **list) {
    if (!o) return;
    if (o->kind != KIT_TRIANGLE) o->box = transformed_box(o->box, o->T);
    o->insert_id = g_inserted++;
    if (!*list) {
        *list = o;
        o->next = NULL;
    } else {
        o->next = (*list)->next;
        (*list)->next = o;
    }
    if (o->isLightSource) {
        if (!light_list) {
            light_list = o;
            o->LSnext = NULL;
        } else {
            o->LSnext = light_list->LSnext;
            light_list->LSnext = o;
        }
    }
}

/* -------------------------------------------------------------- textures */

/* Binary PPM (P6) / PGM (P5) as the reference reads them (utils_path.c:1771-1930): magic line, any number of
 * '#' lines, "w h", a maxval line, then raw bytes.  PPM samples are divided by the file's maxval, PGM samples
 * by 255.  The bytes are kept as read; only the flat scene's consumers turn them into numbers. */
static struct image *read_pnm(const char *filename, const char *magic, int layers, unsigned char **bytes_out,
                              double *divisor_out) {
    FILE *f = fopen(filename, "rb");
    char line[1024];
    if (!f) {
        fprintf(stderr, "Unable to open file %s for reading, please check name and path\n", filename);
        return NULL;
    }
    if (!fgets(line, 1000, f) || strcmp(line, magic) != 0) {
        fprintf(stderr, "Wrong file format, not a %s file or header end-of-line characters missing\n",
                layers == 3 ? ".ppm" : ".pgm");
        fclose(f);
        return NULL;
    }
    do {
        if (!fgets(line, 511, f)) { fclose(f); return NULL; }
    } while (line[0] == '#');
    int w = 0, h = 0;
    if (sscanf(line, "%d %d", &w, &h) != 2 || w <= 0 || h <= 0) { fclose(f); return NULL; }
    if (!fgets(line, 9, f)) { fclose(f); return NULL; }
    int maxval = atoi(line);
    size_t n = (size_t)w * h * layers;
    unsigned char *buf = (unsigned char *)calloc(n, 1);
    struct image *im = (struct image *)calloc(1, sizeof(*im));
    if (!buf || !im) {
        fprintf(stderr, "Out of memory allocating space for image\n");
        free(buf);
        free(im);
        fclose(f);
        return NULL;
    }
    size_t got = fread(buf, 1, n, f);
    (void)got; /* a short file leaves zeros, as calloc + fread does in the reference */
    fclose(f);
    im->sx = w;
    im->sy = h;
    im->rgbdata = NULL;
    *bytes_out = buf;
    *divisor_out = layers == 3 ? (double)maxval : 255.0;
    return im;
}
static unsigned char *g_last_bytes;
static double g_last_divisor;
struct image *readPPMimage(const char *filename) { return read_pnm(filename, "P6\n", 3, &g_last_bytes, &g_last_divisor); }
struct image *readPGMimage(const char *filename) { return read_pnm(filename, "P5\n", 1, &g_last_bytes, &g_last_divisor); }

/* loadTexture (utils_path.c:1271-1320): type 1 texture, 2 normal map, 3 alpha map; images are shared between
 * objects by exact file name; a file that cannot be read leaves the object untextured. */
void loadTexture(struct object3D *o, const char *filename, int type, struct textureNode **t_list) {
    if (!o) return;
    struct image *im = NULL;
    for (struct textureNode *p = *t_list; p; p = p->next)
        if (strcmp(p->name, filename) == 0) { im = p->im; break; }
    if (!im) {
        if (type == 1 || type == 2) im = readPPMimage(filename);
        else if (type == 3) im = readPGMimage(filename);
        if (!im) return;
        struct textureNode *node = (struct textureNode *)calloc(1, sizeof(*node));
        strncpy(node->name, filename, sizeof(node->name) - 1);
        node->type = type;
        node->im = im;
        node->bytes = g_last_bytes;
        node->divisor = g_last_divisor;
        if (!*t_list) *t_list = node;
        else {
            node->next = (*t_list)->next;
            (*t_list)->next = node;
        }
    }
    if (type == 1) o->texImg = im;
    else if (type == 2) o->normalImg = im;
    else o->alphaImg = im;
}

/* ---------------------------------------------------------------- meshes */

/* loadObj (meshes.c:23-101): reads "v x y z" and "f a b c" lines (1-based indices); every face becomes one
 * triangle object with its vertices pre-multiplied by T and the given material.  The reference reads the file in
 * four consecutive sweeps (v, vt, vn, f) without rewinding between them, which is reproduced by the running line
 * cursor.  Beyond what the reference reads: a line may be of any length (the reference's 128-byte buffer cuts a
 * longer one into pieces and reads each piece as a line of its own), a face may name its vertices as a/t/n, and
 * a negative index counts back from the last vertex, as the OBJ format defines it (the reference reads out of
 * bounds on one). */
void loadObj(const char *fname, double diff, double refl, double tran, double r, double g, double b, double refl_sig,
             double r_index, struct object3D **list, double T[4][4]) {
    FILE *f = fopen(fname, "r");
    if (!f) {
        fprintf(stderr, "Unable to open mesh file %s\n", fname);
        return;
    }
    /* every line is kept as its first 127 characters: all a "v" line or the three indices of an "f" line need */
    size_t n_lines = 0, cap = 1024;
    char(*lines)[128] = malloc(cap * 128);
    char *whole = NULL;
    size_t whole_cap = 0;
    while (getline(&whole, &whole_cap, f) >= 0) {
        if (n_lines == cap) { cap *= 2; lines = realloc(lines, cap * 128); }
        strncpy(lines[n_lines], whole, 127);
        lines[n_lines++][127] = 0;
    }
    free(whole);
    fclose(f);
    int vcount = 0, vtcount = 0, vncount = 0, fcount = 0;
    for (size_t k = 0; k < n_lines; k++) {
        if (!strncmp(lines[k], "v ", 2)) vcount++;
        if (!strncmp(lines[k], "vt ", 3)) vtcount++;
        if (!strncmp(lines[k], "vn ", 3)) vncount++;
        if (!strncmp(lines[k], "f ", 2)) fcount++;
    }
    fprintf(stderr, "vertices: %d\n vertex normals: %d\n faces: %d\n", vcount, vncount, fcount);
    struct point3D *vs = (struct point3D *)calloc((size_t)vcount + 1, sizeof(*vs));
    size_t cur = 0;
    int i = 0;
    while (i < vcount && cur < n_lines) {
        const char *ln = lines[cur++];
        if (strncmp(ln, "v ", 2)) continue;
        sscanf(ln, "v %lf %lf %lf", &vs[i].px, &vs[i].py, &vs[i].pz);
        vs[i].pw = 1;
        i++;
    }
    for (i = 0; i < vtcount && cur < n_lines;) /* texture coordinates are parsed and dropped */
        if (!strncmp(lines[cur++], "vt ", 3)) i++;
    for (i = 0; i < vncount && cur < n_lines;) /* vertex normals likewise */
        if (!strncmp(lines[cur++], "vn ", 3)) i++;
    i = 0;
    while (i < fcount && cur < n_lines) {
        const char *ln = lines[cur++];
        if (strncmp(ln, "f ", 2)) continue;
        /* "f a b c" as the reference reads it; "f a/t/n b/t/n c/t/n" (which its sscanf stops at) is accepted too:
         * the vertex index is the number before the first slash */
        int vi[3] = {1, 1, 1};
        const char *q0 = ln + 2;
        for (int q = 0; q < 3; q++) {
            char *end;
            long v = strtol(q0, &end, 10);
            if (end == q0) break;
            vi[q] = (int)v;
            while (*end && *end != ' ' && *end != '\t' && *end != '\n') end++;
            q0 = end;
        }
        struct point3D p[3];
        for (int q = 0; q < 3; q++) {
            int idx = vi[q] < 0 ? vcount + vi[q] : vi[q] - 1; /* -1 is the last vertex */
            if (idx < 0 || idx >= vcount) idx = 0; /* the reference reads out of bounds here */
            p[q] = vs[idx];
            matVecMult(T, &p[q]);
        }
        insertObject(newTriangle(p[0], p[1], p[2], diff, refl, tran, r, g, b, refl_sig, r_index), list);
        i++;
    }
    printf("done mesh\n");
    free(vs);
    free(lines);
}

/* ---------------------------------------------------------------- camera */

/* setupView (utils_path.c:1657-1765): w = -g/|g|, u = w x up, v = u x w, C2W = [u v w e]. */
struct view *setupView(struct point3D *e, struct point3D *g, struct point3D *up, double f, double wl, double wt,
                       double wsize, double fd) {
    struct view *c = (struct view *)calloc(1, sizeof(*c));
    if (!c) {
        fprintf(stderr, "Out of memory setting up camera model!\n");
        return NULL;
    }
    c->e = *e;
    c->e.pw = 1;
    c->w.px = -g->px;
    c->w.py = -g->py;
    c->w.pz = -g->pz;
    c->w.pw = 1;
    normalize(&c->w);
    struct point3D *u = cross(&c->w, up);
    normalize(u);
    c->u = *u;
    c->u.pw = 1;
    struct point3D *v = cross(&c->u, &c->w);
    normalize(v);
    c->v = *v;
    c->v.pw = 1;
    free(u);
    free(v);
    c->f = f;
    c->focus_distance = fd;
    c->wl = wl;
    c->wt = wt;
    c->wsize = wsize;
    const struct point3D *axes[3] = {&c->u, &c->v, &c->w};
    for (int k = 0; k < 3; k++) {
        c->C2W[0][k] = axes[k]->px;
        c->C2W[1][k] = axes[k]->py;
        c->C2W[2][k] = axes[k]->pz;
        c->C2W[3][k] = 0;
        c->W2C[k][0] = axes[k]->px;
        c->W2C[k][1] = axes[k]->py;
        c->W2C[k][2] = axes[k]->pz;
        c->W2C[3][k] = 0;
    }
    c->C2W[0][3] = c->e.px;
    c->C2W[1][3] = c->e.py;
    c->C2W[2][3] = c->e.pz;
    c->C2W[3][3] = 1;
    c->W2C[0][3] = -dot(&c->u, &c->e);
    c->W2C[1][3] = -dot(&c->v, &c->e);
    c->W2C[2][3] = -dot(&c->w, &c->e);
    c->W2C[3][3] = 1;
    return c;
}

/* ---------------------------------------------------------------- images */

struct image *newImage(int size_x, int size_y) {
    struct image *im = (struct image *)calloc(1, sizeof(*im));
    if (im) {
        im->sx = size_x;
        im->sy = size_y;
        im->rgbdata = calloc((size_t)size_x * size_y * 3, sizeof(double));
        if (im->rgbdata) return im;
        free(im);
    }
    fprintf(stderr, "Unable to allocate memory for new image\n");
    return NULL;
}
void deleteImage(struct image *im) {
    if (!im) return;
    free(im->rgbdata);
    free(im);
}

void cleanup(struct object3D *o_list, struct textureNode *t_list) {
    while (o_list) {
        struct object3D *n = o_list->next;
        free(o_list);
        o_list = n;
    }
    while (t_list) {
        struct textureNode *n = t_list->next;
        free(t_list->bytes);
        deleteImage(t_list->im);
        free(t_list);
        t_list = n;
    }
}
