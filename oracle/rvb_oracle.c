/* TEST INFRASTRUCTURE ONLY — see rvb_oracle.h.
 *
 * Plain-C, brute-force restatement of the reference's per-ray hot path.  Every
 * function names the reference lines it follows.  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * so that each float operation below is one IEEE binary32 operation in the order
 * written (no FMA, no reassociation).
 */
#include "rvb_oracle.h"

/* The part of this file up to raytrace_one is also compiled for the GPU by oracle/gpu_oracle.hip (same source, one thread per
 * ray: a brute-force check of whole full-size runs): RVBO_FN marks those functions, RVBO_KERNEL_PART_ONLY drops the rest. */
#ifndef RVBO_FN
#define RVBO_FN
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
typedef struct { v3 v0, v1, v2; } triverts;      /* reference kernel.cpp:56-60 */
typedef struct { v3 position, direction; } ray_t; /* reference kernel.cpp:17-20 */
typedef struct { uint64_t primitive; float distance; int intersects; } hit_t; /* kernel.cpp:34-38 */

/* ---- OpenCL built-ins, as defined in oracle/ref/ref_builtins.cl ---------- */
static inline RVBO_FN v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline RVBO_FN v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline RVBO_FN v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline RVBO_FN v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline RVBO_FN v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline RVBO_FN float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline RVBO_FN v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline RVBO_FN float v3_length(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline RVBO_FN v3 v3_normalize(v3 a)
{
    float l = v3_length(a);
    if (l == 0.0f)
        return a;
    return v3_make(a.x / l, a.y / l, a.z / l);
}
static inline RVBO_FN float pow_cr(float x, float y) { return (float) pow((double) x, (double) y); }
static inline RVBO_FN float atan2_cr(float y, float x) { return (float) atan2((double) y, (double) x); }
static inline RVBO_FN float degrees_f(float x) { return x * 57.295779513082320877f; }

static inline RVBO_FN v3 load_vertex(const float * vertices, uint64_t i)
{
    return v3_make(vertices[4 * i + 0], vertices[4 * i + 1], vertices[4 * i + 2]);
}

/* reference kernel.cpp:14 — a float constant initialised from a double quotient. */
static const float SECONDS_PER_METER = (float) (1.0 / 340.000000);

/* ---- geometry ------------------------------------------------------------ */

/* reference kernel.cpp:62-88 */
static RVBO_FN float triangle_vert_intersection(const triverts * v, const ray_t * ray)
{
    v3 e0 = v3_sub(v->v1, v->v0);
    v3 e1 = v3_sub(v->v2, v->v0);

    v3 pvec = v3_cross(ray->direction, e1);
    float det = v3_dot(e0, pvec);

    if (-RVBO_EPSILON < det && det < RVBO_EPSILON)
        return 0.0f;

    float invdet = 1.0f / det;
    v3 tvec = v3_sub(ray->position, v->v0);
    float ucomp = invdet * v3_dot(tvec, pvec);

    if (ucomp < 0.0f || 1.0f < ucomp)
        return 0.0f;

    v3 qvec = v3_cross(tvec, e0);
    float vcomp = invdet * v3_dot(ray->direction, qvec);

    if (vcomp < 0.0f || 1.0f < vcomp + ucomp)
        return 0.0f;

    return invdet * v3_dot(e1, qvec);
}

/* reference kernel.cpp:95-107, :121-125 */
static RVBO_FN triverts gather_verts(const RvboTriangle * t, const float * vertices)
{
    triverts v;
    v.v0 = load_vertex(vertices, t->v0);
    v.v1 = load_vertex(vertices, t->v1);
    v.v2 = load_vertex(vertices, t->v2);
    return v;
}

/* reference kernel.cpp:109-116 */
static RVBO_FN v3 triangle_verts_normal(const triverts * t)
{
    v3 e0 = v3_sub(t->v1, t->v0);
    v3 e1 = v3_sub(t->v2, t->v0);
    return v3_normalize(v3_cross(e0, e1));
}

/* reference kernel.cpp:129-133: direction - (normal * 2 * dot(direction, normal)) */
static RVBO_FN v3 reflect(v3 normal, v3 direction)
{
    v3 n2 = v3_scale(normal, 2.0f);
    float d = v3_dot(direction, normal);
    return v3_sub(direction, v3_scale(n2, d));
}

/* reference kernel.cpp:167-192 */
static RVBO_FN hit_t ray_triangle_intersection
(   const ray_t * ray, const RvboTriangle * triangles, uint64_t numtriangles, const float * vertices)
{
    hit_t ret = {0, 0.0f, 0};
    for (uint64_t i = 0; i != numtriangles; ++i)
    {
        triverts v = gather_verts(triangles + i, vertices);
        float distance = triangle_vert_intersection(&v, ray);
        if (distance > RVBO_EPSILON && (!ret.intersects || distance < ret.distance))
        {
            ret.primitive = i;
            ret.distance = distance;
            ret.intersects = 1;
        }
    }
    return ret;
}

/* reference kernel.cpp:194-214: pow(M_E, distance * AIR) * 1 */
static RVBO_FN void attenuation_for_distance(float distance, const float air[8], float out[8])
{
    const float e = (float) 2.7182818284590452354; /* M_E converted to the float8 element type */
    for (int b = 0; b != 8; ++b)
        out[b] = pow_cr(e, distance * air[b]) * 1.0f;
}

/* reference kernel.cpp:216-221: *p += -n * dot(n, *p - t->v0) * 2 */
static RVBO_FN void mirror_point(v3 * p, const triverts * t)
{
    v3 n = triangle_verts_normal(t);
    float d = v3_dot(n, v3_sub(*p, t->v0));
    v3 delta = v3_scale(v3_scale(v3_neg(n), d), 2.0f);
    *p = v3_add(*p, delta);
}

/* reference kernel.cpp:223-229 */
static RVBO_FN void mirror_verts(triverts * in, const triverts * t)
{
    mirror_point(&in->v0, t);
    mirror_point(&in->v1, t);
    mirror_point(&in->v2, t);
}

/* reference kernel.cpp:243-265 */
static RVBO_FN void add_image
(   v3 mic_position, v3 mic_reflection, v3 source,
    RvboImpulse * image_source, uint64_t * image_source_index,
    uint64_t thread_index, uint64_t thread_offset_index,
    const float volume[8], uint64_t object_index, const float air[8])
{
    v3 init_diff = v3_sub(source, mic_reflection);
    float init_dist = v3_length(init_diff);
    uint64_t offset = thread_index * RVBO_NUM_IMAGE_SOURCE + thread_offset_index;
    float att[8];
    attenuation_for_distance(init_dist, air, att);
    RvboImpulse imp;
    memset(&imp, 0, sizeof(imp));
    for (int b = 0; b != 8; ++b)
        imp.volume[b] = volume[b] * att[b];
    v3 pos = v3_add(mic_position, init_diff);
    imp.position[0] = pos.x;
    imp.position[1] = pos.y;
    imp.position[2] = pos.z;
    imp.time = SECONDS_PER_METER * init_dist;
    image_source[offset] = imp;
    image_source_index[offset] = object_index;
}

/* reference kernel.cpp:274-296 */
static RVBO_FN int point_intersection
(   v3 begin, v3 point, const RvboTriangle * triangles, uint64_t numtriangles, const float * vertices)
{
    v3 begin_to_point = v3_sub(point, begin);
    float mag = v3_length(begin_to_point);
    ray_t to_point = {begin, v3_normalize(begin_to_point)};
    hit_t inter = ray_triangle_intersection(&to_point, triangles, numtriangles, vertices);
    return (!inter.intersects) || inter.distance > mag;
}

/* reference kernel.cpp:298-302 */
static RVBO_FN v3 get_direction(v3 from, v3 to) { return v3_normalize(v3_sub(to, from)); }

/* reference kernel.cpp:304-503, one work-item */
static RVBO_FN void raytrace_one
(   uint64_t i, const float * directions, v3 position,
    const RvboTriangle * triangles, uint64_t numtriangles, const float * vertices,
    v3 source, const RvboSurface * surfaces,
    RvboImpulse * impulses, RvboImpulse * image_source, uint64_t * image_source_index,
    uint64_t outputOffset, const float air[8])
{
    ray_t ray = {source, v3_make(directions[4 * i], directions[4 * i + 1], directions[4 * i + 2])};
    float distance = 0.0f;
    float volume[8];
    for (int b = 0; b != 8; ++b)
        volume[b] = 1.0f;

    triverts prev_primitives[RVBO_NUM_IMAGE_SOURCE - 1];
    v3 mic_reflection = position;

    /* kernel.cpp:335-357: direct path */
    if (point_intersection(source, mic_reflection, triangles, numtriangles, vertices))
        add_image(position, mic_reflection, source, image_source, image_source_index, i, 0, volume, 0, air);

    for (uint64_t index = 0; index != outputOffset; ++index)
    {
        /* kernel.cpp:363-375 */
        hit_t closest = ray_triangle_intersection(&ray, triangles, numtriangles, vertices);
        if (!closest.intersects)
            break;

        const RvboTriangle * triangle = triangles + closest.primitive;

        /* kernel.cpp:379-457: image-source validation for the first 9 reflections */
        if (index < RVBO_NUM_IMAGE_SOURCE - 1)
        {
            triverts current = gather_verts(triangle, vertices);
            for (uint64_t k = 0; k != index; ++k)
                mirror_verts(&current, prev_primitives + k);
            prev_primitives[index] = current;

            mirror_point(&mic_reflection, &current);

            v3 dir = get_direction(source, mic_reflection);
            ray_t to_mic = {source, dir};
            int intersects = 1;
            v3 prev_intersection = source;
            for (uint64_t k = 0; k != index + 1 && intersects; ++k)
            {
                float to_intersection = triangle_vert_intersection(prev_primitives + k, &to_mic);
                if (to_intersection <= RVBO_EPSILON)
                {
                    intersects = 0;
                    break;
                }

                v3 intersection_point = v3_add(source, v3_scale(dir, to_intersection));
                for (int64_t l = (int64_t) k - 1; l != -1; --l)
                    mirror_point(&intersection_point, prev_primitives + l);

                ray_t intermediate = {prev_intersection, get_direction(prev_intersection, intersection_point)};
                hit_t inter = ray_triangle_intersection(&intermediate, triangles, numtriangles, vertices);

                v3 nip = v3_add(intermediate.position, v3_scale(intermediate.direction, inter.distance));
                int lo = (nip.x - RVBO_EPSILON < intersection_point.x)
                      && (nip.y - RVBO_EPSILON < intersection_point.y)
                      && (nip.z - RVBO_EPSILON < intersection_point.z);
                int hi = (intersection_point.x < nip.x + RVBO_EPSILON)
                      && (intersection_point.y < nip.y + RVBO_EPSILON)
                      && (intersection_point.z < nip.z + RVBO_EPSILON);
                intersects = inter.intersects && lo && hi;

                prev_intersection = intersection_point;
            }

            if (intersects)
                intersects = point_intersection(prev_intersection, position, triangles, numtriangles, vertices);

            if (intersects)
                add_image(position, mic_reflection, source, image_source, image_source_index,
                          i, index + 1, volume, closest.primitive + 1, air);
        }

        /* kernel.cpp:459-461 */
        v3 intersection = v3_add(ray.position, v3_scale(ray.direction, closest.distance));
        float new_dist = distance + closest.distance;
        const RvboSurface * surface = surfaces + triangle->surface;
        float new_vol[8];
        for (int b = 0; b != 8; ++b)
            new_vol[b] = -volume[b] * surface->specular[b];

        /* kernel.cpp:463-471: diffuse shadow ray to the microphone */
        int is_intersection = point_intersection(intersection, position, triangles, numtriangles, vertices);
        float dist = is_intersection ? new_dist + v3_length(v3_sub(position, intersection)) : 0.0f;

        /* kernel.cpp:478-490 */
        triverts tv = gather_verts(triangle, vertices);
        v3 normal = triangle_verts_normal(&tv);
        float diff = fabsf(v3_dot(normal, ray.direction));

        RvboImpulse imp;
        memset(&imp, 0, sizeof(imp));
        if (is_intersection)
        {
            float att[8];
            attenuation_for_distance(dist, air, att);
            for (int b = 0; b != 8; ++b)
                imp.volume[b] = ((new_vol[b] * att[b]) * surface->diffuse[b]) * diff;
        }
        imp.position[0] = intersection.x;
        imp.position[1] = intersection.y;
        imp.position[2] = intersection.z;
        imp.time = SECONDS_PER_METER * dist;
        impulses[i * outputOffset + index] = imp;

        /* kernel.cpp:492-501 */
        ray.position = intersection;
        ray.direction = reflect(normal, ray.direction);
        distance = new_dist;
        memcpy(volume, new_vol, sizeof(volume));
    }
}

#ifndef RVBO_KERNEL_PART_ONLY
void rvbo_raytrace
(   const float * directions, uint64_t nrays,
    const RvboTriangle * triangles, uint64_t ntriangles,
    const float * vertices,
    const RvboSurface * surfaces,
    const float mic[3], const float source[3],
    uint64_t nreflections, const float air[8],
    RvboImpulse * impulses, RvboImpulse * image_source, uint64_t * image_source_index,
    int nthreads)
{
    /* reference rayverb.cpp:600-616: zero-filled outputs */
    memset(impulses, 0, sizeof(RvboImpulse) * nrays * nreflections);
    memset(image_source, 0, sizeof(RvboImpulse) * nrays * RVBO_NUM_IMAGE_SOURCE);
    memset(image_source_index, 0, sizeof(uint64_t) * nrays * RVBO_NUM_IMAGE_SOURCE);

    v3 position = v3_make(mic[0], mic[1], mic[2]);
    v3 src = v3_make(source[0], source[1], source[2]);
#ifdef _OPENMP
    if (nthreads <= 0)
        nthreads = omp_get_num_procs();
#endif
    (void) nthreads;
    #pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t) nrays; ++i)
        raytrace_one((uint64_t) i, directions, position, triangles, ntriangles, vertices, src, surfaces,
                     impulses, image_source, image_source_index, nreflections, air);
}

int rvbo_closest_hit
(   const float origin[3], const float direction[3],
    const RvboTriangle * triangles, uint64_t ntriangles, const float * vertices,
    uint64_t * primitive, float * distance)
{
    ray_t r = {v3_make(origin[0], origin[1], origin[2]), v3_make(direction[0], direction[1], direction[2])};
    hit_t h = ray_triangle_intersection(&r, triangles, ntriangles, vertices);
    *primitive = h.primitive;
    *distance = h.distance;
    return h.intersects;
}

int rvbo_point_visible
(   const float begin[3], const float point[3],
    const RvboTriangle * triangles, uint64_t ntriangles, const float * vertices)
{
    return point_intersection(v3_make(begin[0], begin[1], begin[2]), v3_make(point[0], point[1], point[2]),
                              triangles, ntriangles, vertices);
}

/* ---- host de-dup: reference rayverb.cpp:654-676, then :692-706 ------------ */

typedef struct { uint64_t key[RVBO_NUM_IMAGE_SOURCE]; int len; RvboImpulse value; } tally_entry;

/* std::map<std::vector<unsigned long>, Impulse> orders keys lexicographically,
 * a proper prefix before the longer key. */
static int key_compare(const uint64_t * a, int la, const uint64_t * b, int lb)
{
    int n = la < lb ? la : lb;
    for (int i = 0; i != n; ++i)
    {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return (la > lb) - (la < lb);
}

static int tally_qsort_cmp(const void * pa, const void * pb)
{
    const tally_entry * a = (const tally_entry *) pa;
    const tally_entry * b = (const tally_entry *) pb;
    return key_compare(a->key, a->len, b->key, b->len);
}

uint64_t rvbo_collect_images
(   const RvboImpulse * image_source, const uint64_t * image_source_index, uint64_t nrays,
    int remove_direct, RvboImpulse * out, uint64_t max_out)
{
    /* Small open-addressing table keyed on the whole index vector; insert-if-absent in ray order. */
    uint64_t cap = 1024;
    while (cap < nrays * 4 + 16)
        cap <<= 1;
    tally_entry * table = (tally_entry *) calloc(cap, sizeof(tally_entry));
    uint64_t count = 0;

    for (uint64_t j = 0; j != nrays * RVBO_NUM_IMAGE_SOURCE; j += RVBO_NUM_IMAGE_SOURCE)
    {
        for (int k = 1; k != RVBO_NUM_IMAGE_SOURCE + 1; ++k)
        {
            const uint64_t * key = image_source_index + j;
            if (!(k == 1 || key[k - 1] != 0))
                continue;
            uint64_t h = 1469598103934665603ull;
            for (int m = 0; m != k; ++m)
                h = (h ^ key[m]) * 1099511628211ull;
            h ^= (uint64_t) k * 0x9E3779B97F4A7C15ull;
            uint64_t slot = h & (cap - 1);
            for (;;)
            {
                tally_entry * e = table + slot;
                if (e->len == 0)
                {
                    memcpy(e->key, key, sizeof(uint64_t) * (size_t) k);
                    e->len = k;
                    e->value = image_source[j + (uint64_t) k - 1];
                    ++count;
                    break;
                }
                if (key_compare(e->key, e->len, key, k) == 0)
                    break; /* first ray wins */
                slot = (slot + 1) & (cap - 1);
            }
            if (count * 2 > cap)
            {   /* grow */
                uint64_t ncap = cap << 1;
                tally_entry * nt = (tally_entry *) calloc(ncap, sizeof(tally_entry));
                for (uint64_t s = 0; s != cap; ++s)
                {
                    if (table[s].len == 0) continue;
                    uint64_t hh = 1469598103934665603ull;
                    for (int m = 0; m != table[s].len; ++m)
                        hh = (hh ^ table[s].key[m]) * 1099511628211ull;
                    hh ^= (uint64_t) table[s].len * 0x9E3779B97F4A7C15ull;
                    uint64_t ns = hh & (ncap - 1);
                    while (nt[ns].len != 0) ns = (ns + 1) & (ncap - 1);
                    nt[ns] = table[s];
                }
                free(table);
                table = nt;
                cap = ncap;
            }
        }
    }

    tally_entry * flat = (tally_entry *) malloc(sizeof(tally_entry) * (count ? count : 1));
    uint64_t n = 0;
    for (uint64_t s = 0; s != cap; ++s)
        if (table[s].len != 0)
            flat[n++] = table[s];
    free(table);
    qsort(flat, n, sizeof(tally_entry), tally_qsort_cmp);

    uint64_t written = 0;
    for (uint64_t s = 0; s != n; ++s)
    {
        if (remove_direct && flat[s].len == 1 && flat[s].key[0] == 0)
            continue; /* rayverb.cpp:695-696: temp.erase({0}) */
        if (written < max_out)
            out[written] = flat[s].value;
        ++written;
    }
    free(flat);
    return written;
}

/* ---- attenuation ----------------------------------------------------------- */

static int any_nonzero(const float v[8])
{
    for (int b = 0; b != 8; ++b)
        if (v[b] != 0.0f)
            return 1;
    return 0;
}

/* reference kernel.cpp:505-535 */
void rvbo_attenuate_speaker
(   const float mic[3], const RvboImpulse * in, uint64_t n,
    const float direction[3], float coefficient, RvboAttenuated * out)
{
    v3 mic_pos = v3_make(mic[0], mic[1], mic[2]);
    v3 sdir = v3_make(direction[0], direction[1], direction[2]);
    for (uint64_t i = 0; i != n; ++i)
    {
        RvboAttenuated o;
        memset(&o, 0, sizeof(o));
        if (any_nonzero(in[i].volume))
        {
            v3 pos = v3_make(in[i].position[0], in[i].position[1], in[i].position[2]);
            v3 d = get_direction(mic_pos, pos);
            float attenuation = (1 - coefficient) + coefficient * v3_dot(v3_normalize(d), v3_normalize(sdir));
            for (int b = 0; b != 8; ++b)
                o.volume[b] = in[i].volume[b] * attenuation;
            o.time = in[i].time;
        }
        out[i] = o;
    }
}

/* reference kernel.cpp:537-549 */
static v3 transform(v3 pointing, v3 up, v3 d)
{
    v3 x = v3_normalize(v3_cross(up, pointing));
    v3 y = v3_cross(pointing, x);
    v3 z = pointing;
    return v3_make(v3_dot(x, d), v3_dot(y, d), v3_dot(z, d));
}

/* reference kernel.cpp:551-584 */
static int64_t hrtf_index(v3 pointing, v3 up, v3 impulse_direction)
{
    v3 t = transform(pointing, up, impulse_direction);
    float az = atan2_cr(t.x, t.z);
    float el = atan2_cr(t.y, sqrtf(t.x * t.x + t.z * t.z));
    int64_t a = (int64_t) (degrees_f(az) + 180);
    a %= 360;
    int64_t e = (int64_t) degrees_f(el);
    e = 90 - e;
    return a * 180 + e;
}

int64_t rvbo_hrtf_index(const float pointing[3], const float up[3], const float direction[3])
{
    return hrtf_index(v3_make(pointing[0], pointing[1], pointing[2]), v3_make(up[0], up[1], up[2]),
                      v3_make(direction[0], direction[1], direction[2]));
}

/* reference kernel.cpp:586-625 */
void rvbo_attenuate_hrtf
(   const float mic[3], const RvboImpulse * in, uint64_t n,
    const float * table, const float pointing_[3], const float up_[3],
    uint64_t channel, RvboAttenuated * out)
{
    const float width = 0.1f; /* const float WIDTH = 0.1 */
    v3 mic_pos = v3_make(mic[0], mic[1], mic[2]);
    v3 pointing = v3_make(pointing_[0], pointing_[1], pointing_[2]);
    v3 up = v3_make(up_[0], up_[1], up_[2]);
    v3 ear_pos = v3_add(transform(pointing, up, v3_make(channel == 0 ? -width : width, 0.0f, 0.0f)), mic_pos);

    for (uint64_t i = 0; i != n; ++i)
    {
        RvboAttenuated o;
        memset(&o, 0, sizeof(o));
        if (any_nonzero(in[i].volume))
        {
            v3 pos = v3_make(in[i].position[0], in[i].position[1], in[i].position[2]);
            int64_t idx = hrtf_index(pointing, up, get_direction(mic_pos, pos));
            const float * att = table + 8 * idx;
            float dist0 = v3_length(v3_sub(pos, mic_pos));
            float dist1 = v3_length(v3_sub(pos, ear_pos));
            float diff = dist1 - dist0;
            for (int b = 0; b != 8; ++b)
                o.volume[b] = in[i].volume[b] * att[b];
            o.time = in[i].time + diff * SECONDS_PER_METER;
        }
        out[i] = o;
    }
}

/* ---- predelay + time binning ---------------------------------------------- */

/* reference rayverb.h:49-97: earliest non-zero time over every impulse of every channel */
float rvbo_find_predelay(const RvboAttenuated * const * channels, uint64_t nchannels, uint64_t n)
{
    float a = 0.0f;
    for (uint64_t c = 0; c != nchannels; ++c)
        for (uint64_t i = 0; i != n; ++i)
        {
            float pd = channels[c][i].time;
            if (a == 0.0f)
                a = pd;
            else if (pd != 0.0f)
                a = a < pd ? a : pd;
        }
    return a;
}

void rvbo_fix_predelay(RvboAttenuated * impulses, uint64_t n, float seconds)
{
    for (uint64_t i = 0; i != n; ++i)
        impulses[i].time = impulses[i].time > seconds ? impulses[i].time - seconds : 0.0f;
}

/* reference rayverb.cpp:53-57 */
uint64_t rvbo_flatten_bins(const RvboAttenuated * impulses, uint64_t n, float samplerate)
{
    float maxtime = 0.0f;
    for (uint64_t i = 0; i != n; ++i)
        maxtime = maxtime > impulses[i].time ? maxtime : impulses[i].time;
    return (uint64_t) (roundf(maxtime * samplerate) + 1);
}

/* reference rayverb.cpp:59-76 */
void rvbo_flatten(const RvboAttenuated * impulses, uint64_t n, float samplerate, float * out, uint64_t nbins)
{
    memset(out, 0, sizeof(float) * 8 * nbins);
    for (uint64_t i = 0; i != n; ++i)
    {
        uint64_t sample = (uint64_t) roundf(impulses[i].time * samplerate);
        for (int b = 0; b != 8; ++b)
            out[(uint64_t) b * nbins + sample] += impulses[i].volume[b];
    }
}

#endif /* RVBO_KERNEL_PART_ONLY */
