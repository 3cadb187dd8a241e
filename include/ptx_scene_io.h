/*
 * ptx_scene_io.h -- write / read a ptx_scene_desc as one binary file.
 *
 * File layout (little endian, native struct layout of ptx_scene.h):
 *   u64 magic | i32 n_objects n_nodes n_lights n_images | u64 n_texel_bytes
 *   ptx_camera | ptx_object[n_objects] | ptx_node[n_nodes]
 *   i32 lights[n_lights] | ptx_image[n_images] | u8 texels[n_texel_bytes]
 *
 * Header-only so that the reference dump tool, the CPU oracle and the host
 * library can share it without a link dependency.
 */
#ifndef PTX_SCENE_IO_H
#define PTX_SCENE_IO_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ptx_scene.h"

static inline int ptx_scene_write(const char *path, const ptx_scene_desc *s) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    uint64_t magic = PTX_SCENE_MAGIC;
    int ok = 1;
    ok &= fwrite(&magic, 8, 1, f) == 1;
    ok &= fwrite(&s->n_objects, 4, 1, f) == 1;
    ok &= fwrite(&s->n_nodes, 4, 1, f) == 1;
    ok &= fwrite(&s->n_lights, 4, 1, f) == 1;
    ok &= fwrite(&s->n_images, 4, 1, f) == 1;
    ok &= fwrite(&s->n_texel_bytes, 8, 1, f) == 1;
    ok &= fwrite(&s->cam, sizeof(ptx_camera), 1, f) == 1;
    if (s->n_objects) ok &= fwrite(s->objects, sizeof(ptx_object), s->n_objects, f) == (size_t)s->n_objects;
    if (s->n_nodes) ok &= fwrite(s->nodes, sizeof(ptx_node), s->n_nodes, f) == (size_t)s->n_nodes;
    if (s->n_lights) ok &= fwrite(s->lights, 4, s->n_lights, f) == (size_t)s->n_lights;
    if (s->n_images) ok &= fwrite(s->images, sizeof(ptx_image), s->n_images, f) == (size_t)s->n_images;
    if (s->n_texel_bytes) ok &= fwrite(s->texels, 1, s->n_texel_bytes, f) == s->n_texel_bytes;
    fclose(f);
    return ok ? 0 : -2;
}

/* Reads a scene file into freshly malloc'd arrays owned by the caller
 * (release with ptx_scene_free). */
static inline int ptx_scene_read(const char *path, ptx_scene_desc *s) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    memset(s, 0, sizeof(*s));
    int ok = 1;
    ok &= fread(&s->magic, 8, 1, f) == 1;
    if (!ok || s->magic != PTX_SCENE_MAGIC) { fclose(f); return -3; }
    ok &= fread(&s->n_objects, 4, 1, f) == 1;
    ok &= fread(&s->n_nodes, 4, 1, f) == 1;
    ok &= fread(&s->n_lights, 4, 1, f) == 1;
    ok &= fread(&s->n_images, 4, 1, f) == 1;
    ok &= fread(&s->n_texel_bytes, 8, 1, f) == 1;
    ok &= fread(&s->cam, sizeof(ptx_camera), 1, f) == 1;
    if (!ok || s->n_objects < 0 || s->n_nodes < 0 || s->n_lights < 0 || s->n_images < 0) {
        fclose(f);
        return -2;
    }
    ptx_object *objs = (ptx_object *)malloc(sizeof(ptx_object) * (size_t)(s->n_objects + 1));
    ptx_node *nodes = (ptx_node *)malloc(sizeof(ptx_node) * (size_t)(s->n_nodes + 1));
    int32_t *lights = (int32_t *)malloc(4 * (size_t)(s->n_lights + 1));
    ptx_image *images = (ptx_image *)malloc(sizeof(ptx_image) * (size_t)(s->n_images + 1));
    uint8_t *texels = (uint8_t *)malloc(s->n_texel_bytes + 1);
    if (!objs || !nodes || !lights || !images || !texels) { fclose(f); return -4; }
    if (s->n_objects) ok &= fread(objs, sizeof(ptx_object), s->n_objects, f) == (size_t)s->n_objects;
    if (s->n_nodes) ok &= fread(nodes, sizeof(ptx_node), s->n_nodes, f) == (size_t)s->n_nodes;
    if (s->n_lights) ok &= fread(lights, 4, s->n_lights, f) == (size_t)s->n_lights;
    if (s->n_images) ok &= fread(images, sizeof(ptx_image), s->n_images, f) == (size_t)s->n_images;
    if (s->n_texel_bytes) ok &= fread(texels, 1, s->n_texel_bytes, f) == s->n_texel_bytes;
    fclose(f);
    s->objects = objs;
    s->nodes = nodes;
    s->lights = lights;
    s->images = images;
    s->texels = texels;
    return ok ? 0 : -2;
}

static inline void ptx_scene_free(ptx_scene_desc *s) {
    free((void *)s->objects);
    free((void *)s->nodes);
    free((void *)s->lights);
    free((void *)s->images);
    free((void *)s->texels);
    memset(s, 0, sizeof(*s));
}

#endif
