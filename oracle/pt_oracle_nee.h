/*
 * TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (included by pt_oracle.c).
 *
 * Direct-light sampling: restatement of the code the reference carries but never calls from
 * Trace (SampleLight scene.h:150-170, Pdf scene.h:110-144, IntersectP scene.h:101-108,
 * Primitive::Sample primitive.h:50-72, VisibilityTester light.h:23-32, UniformSampleSphere /
 * UniformSampleCone / UniformConePdf / PowerHeuristic montecarlo.h:135-159, CoordinateSystem
 * geometry.h:548-558, Material::F material.h:37-43), and the opt-in integrator built on it.
 *
 * Three of those functions cannot work as written; each exists here twice:
 *   as_written = 1  the reference's text, bug for bug (pinned bit-exactly against the
 *                   reference's host build by tests/golden/nee_components.npz);
 *   as_written = 0  the repaired form the integrator uses:
 *     (1) Primitive::Sample(p, ...) primitive.h:67  `if(thit = Intersect(r) > 0.f)` assigns the
 *         comparison (0 or 1), so the sampled "surface point" is the foot of the centre on the
 *         ray, inside the sphere, and every shadow ray towards it is blocked by the light's own
 *         surface.  Repaired as in pbrt's Sphere::Sample, which the code transcribes: thit is
 *         the intersection distance, the projection only when the ray misses.
 *     (2) SampleLight scene.h:167 evaluates light->L(p, -wi, n): Light::L (light.h:43-45) tests
 *         n.Dot(-w) > 0 with w the direction of travel (Trace passes the ray direction,
 *         pathtracer.cu:136), so passing -wi asks for the back face and returns black for every
 *         point outside the light.  Repaired: L(sp, wi, n).
 *     (3) Pdf(scene, prim, p, wi) scene.h:114-124 (p on or inside the light): assigns the bool of
 *         the scene-wide Intersect to thit and uses the normal at p.  Repaired as pbrt's
 *         Shape::Pdf: hit distance and normal of this primitive's own intersection.
 * A fourth deviation concerns the integrator only (see trace_nee): occlusion of an area light is
 * decided by the nearest hit, not by VisibilityTester's shortened segment.
 */

/* UniformSampleSphere, montecarlo.h:135-142 */
static v3 uniform_sample_sphere(float u1, float u2)
{
    float z = 1.f - 2.f * u1;
    float e = 1.f - z * z;
    float r = sqrtf(0.f > e ? 0.f : e);
    float phi = 2.f * PT_M_PI * u2;
    float s, c;
    ptor_sincosf(phi, &s, &c);
    return v3_make(r * c, r * s, z);
}

/* Lerp, globals.h:103-105 */
static float lerpf(float t, float v1, float v2) { return (1.f - t) * v1 + t * v2; }

/* UniformSampleCone, montecarlo.h:144-150 */
static v3 uniform_sample_cone(float u1, float u2, float costhetamax, v3 x, v3 y, v3 z)
{
    float costheta = lerpf(u1, costhetamax, 1.f);
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = u2 * 2.f * PT_M_PI;
    float s, c;
    ptor_sincosf(phi, &s, &c);
    return v3_add(v3_add(v3_scale(v3_scale(x, c), sintheta), v3_scale(v3_scale(y, s), sintheta)),
                  v3_scale(z, costheta));
}

/* UniformConePdf, montecarlo.h:152-154 */
static float uniform_cone_pdf(float cos_theta_max) { return 1.f / (2.f * PT_M_PI * (1.f - cos_theta_max)); }

/* PowerHeuristic, montecarlo.h:156-159 */
static float power_heuristic(int nf, float f_pdf, int ng, float g_pdf)
{
    float f = nf * f_pdf, g = ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

/* CoordinateSystem, geometry.h:548-558 */
static void coordinate_system(v3 v1, v3 *v2, v3 *v3o)
{
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float inv_len = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = v3_make(-v1.z * inv_len, 0.f, v1.x * inv_len);
    } else {
        float inv_len = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = v3_make(0.f, v1.z * inv_len, -v1.y * inv_len);
    }
    *v3o = v3_cross(v1, *v2);
}

/* Material::F, material.h:37-43 */
static v3 material_f(const ptor_material *mat)
{
    v3 f = v3_make(0.f, 0.f, 0.f);
    if (mat->type == 0) f = v3_add(f, v3_scale(v3_load(mat->color), PT_INV_PI));
    return f;
}

/* Pdf(mat, wo, wi, n), scene.h:136-144 */
static float material_pdf(const ptor_material *mat, v3 wo, v3 wi, v3 n)
{
    if (mat->type == 0) return -v3_dot(wo, wi) > 0.f ? v3_dot(wi, n) * PT_INV_PI : 0.f;
    return 0.f;
}

/* Light::L, light.h:43-45 */
static v3 light_L(const ptor_light *light, v3 w, v3 n)
{
    return v3_dot(n, v3_neg(w)) > 0.f ? v3_load(light->intensity) : v3_make(0.f, 0.f, 0.f);
}

typedef struct { v3 o, d; float mint, maxt; } segment_t;

/* VisibilityTester::SetSegment, light.h:24-29 */
static segment_t visibility_segment(v3 p1, float eps1, v3 p2, float eps2)
{
    segment_t s;
    float dist = v3_length(v3_sub(p1, p2));
    s.mint = eps1;
    s.maxt = dist * (1.f - eps2);
    s.o = p1;
    s.d = v3_div(v3_sub(p2, p1), dist);
    return s;
}

/* Primitive::Sample(sample, ns), primitive.h:50-54 */
static v3 sphere_sample_area(const ptor_sphere *s, float u, float v, v3 *ns)
{
    v3 pos = v3_load(s->pos);
    v3 p = v3_add(pos, v3_scale(uniform_sample_sphere(u, v), s->radius));
    *ns = v3_div(v3_sub(p, pos), s->radius);
    return p;
}

/* Primitive::Sample(p, sample, ns), primitive.h:55-72 */
static v3 sphere_sample_from(const ptor_sphere *s, v3 p, float u, float v, v3 *ns, int as_written)
{
    v3 pos = v3_load(s->pos);
    v3 wc = v3_normalize(v3_sub(pos, p));
    v3 wcx, wcy;
    coordinate_system(wc, &wcx, &wcy);
    v3 dp = v3_sub(p, pos);
    float dist2 = v3_dot(dp, dp);
    if (dist2 - s->radius * s->radius < 1e-4f) return sphere_sample_area(s, u, v, ns);
    float sin_theta_max2 = s->radius * s->radius / dist2;
    float e = 1.f - sin_theta_max2;
    float cos_theta_max = sqrtf(0.f > e ? 0.f : e);
    v3 d = uniform_sample_cone(u, v, cos_theta_max, wcx, wcy, wc);
    float thit = sphere_intersect(s, p, d, PT_EPSILON, PT_INF);
    if (as_written) {
        thit = (float)(thit > 0.f);                                  /* primitive.h:67 */
        if (thit) {
            d = v3_normalize(d);                                     /* r.d.Normalize() works in place */
            thit = v3_dot(v3_sub(pos, p), d);
        }
    } else {
        if (!(thit > 0.f)) thit = v3_dot(v3_sub(pos, p), v3_normalize(d));
    }
    v3 ps = v3_add(p, v3_scale(d, thit));
    *ns = v3_normalize(v3_sub(ps, pos));
    return ps;
}

/* Primitive::Area, primitive.h:73 */
static float sphere_area(const ptor_sphere *s) { return 4 * PT_M_PI * s->radius * s->radius; }

/* Pdf(scene, prim, p, wi), scene.h:110-130 */
static float sphere_light_pdf(const scene_ctx *sc, const ptor_sphere *s, v3 p, v3 wi, int as_written)
{
    float rad = s->radius;
    v3 pos = v3_load(s->pos);
    v3 dp = v3_sub(p, pos);
    float dist2 = v3_dot(dp, dp);
    if (dist2 - rad * rad < 1e-4f) {
        float thit;
        v3 n;
        if (as_written) {
            isect_t is;
            thit = (float)scene_intersect(sc, p, wi, &is);           /* scene.h:117: bool -> float */
            if (thit < 0.f) return 0.f;
            n = v3_div(v3_sub(p, pos), rad);
        } else {
            thit = sphere_intersect(s, p, wi, PT_EPSILON, PT_INF);
            if (thit < 0.f) return 0.f;
            n = v3_div(v3_sub(v3_add(p, v3_scale(wi, thit)), pos), rad);
        }
        v3 dq = v3_sub(p, v3_add(p, v3_scale(wi, thit)));
        float pdf = v3_dot(dq, dq) / (fabsf(v3_dot(n, v3_neg(wi))) * sphere_area(s));
        if (isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    float sin_theta_max2 = rad * rad / dist2;
    float e = 1.f - sin_theta_max2;
    float cos_theta_max = sqrtf(0.f > e ? 0.f : e);
    return uniform_cone_pdf(cos_theta_max);
}

/* IntersectP, scene.h:101-108 with Primitive::IntersectP, primitive.h:46-49 */
static int scene_intersect_p(const scene_ctx *sc, v3 o, v3 d, float tmin, float tmax)
{
    for (uint32_t i = 0; i < sc->n_prims; ++i)
        if (sphere_intersect(&sc->prims[i], o, d, tmin, tmax) > 0.f) return 1;
    return 0;
}

/* SampleLight, scene.h:150-170 */
static v3 sample_light(const scene_ctx *sc, const ptor_light *light, v3 p, float u, float v, v3 *wi, float *pdf,
                       segment_t *vt, int as_written)
{
    if (light->type == 0) {
        v3 pos = v3_load(light->u.pos);
        *wi = v3_normalize(v3_sub(pos, p));
        *pdf = 1.f;
        *vt = visibility_segment(p, PT_EPSILON, pos, 0.f);
        v3 dp = v3_sub(pos, p);
        return v3_div(v3_load(light->intensity), v3_dot(dp, dp));
    } else if (light->type == 1) {
        v3 n;
        const ptor_sphere *prim = &sc->prims[light->u.prim_id];
        v3 sp = sphere_sample_from(prim, p, u, v, &n, as_written);
        *wi = v3_normalize(v3_sub(sp, p));
        *pdf = sphere_light_pdf(sc, prim, p, *wi, as_written);
        *vt = visibility_segment(p, PT_EPSILON, sp, PT_EPSILON);
        return light_L(light, as_written ? v3_neg(*wi) : *wi, n);
    }
    return v3_make(0.f, 0.f, 0.f);
}

static int v3_is_black(v3 c) { return c.x == 0.f && c.y == 0.f && c.z == 0.f; }

/*
 * Trace (pathtracer.cu:112-170) with next-event estimation and multiple importance sampling --
 * the integrator the commented-out lines pathtracer.cu:117,133,148 prepare for, built as pbrt's
 * PathIntegrator / EstimateDirect, with one BSDF sample serving both the continuation and the
 * BSDF-side of the MIS pair:
 *   - emission found by a camera ray or after a specular / transmissive bounce counts in full
 *     (pathtracer.cu:133 as commented); emission found after a diffuse bounce is weighted by
 *     PowerHeuristic(bsdfPdf, lightPdf / nLights);
 *   - at every diffuse vertex one light is chosen uniformly and sampled (SampleLight), weighted
 *     by PowerHeuristic(lightPdf / nLights, bsdfPdf), if the light is visible.  For an area light
 *     "visible" means that the nearest hit along wi is the light's own sphere -- the same rule by
 *     which a BSDF-sampled ray finds emission -- and not VisibilityTester's segment test:
 *     SetSegment(p, EPSILON, sp, EPSILON) stops the segment 3 % (EPSILON = 3e-2, relative) short
 *     of the light, so the demo scene's ceiling, 0.4 units in front of most of the light sphere,
 *     is not seen as an occluder and the image comes out 2.1x too bright.  Point lights keep
 *     the segment test (their eps2 is 0);
 *     directions the BSDF sampler would have rejected (material Pdf <= 0: scene.h:139 and :184)
 *     count nothing, so that the expectation equals that of Trace itself.
 * Random numbers: path stream as in Trace; light stream slot `bounces` = (u, v, choice, -).
 * `hits` records the nearest-hit calls only; n_rays counts shadow rays as well.
 */
static v3 trace_nee(const scene_ctx *sc, v3 o, v3 d, uint32_t pixel, uint32_t iteration,
                    int32_t *hits, uint64_t *n_rays)
{
    v3 T = v3_make(1.f, 1.f, 1.f), L = v3_make(0.f, 0.f, 0.f);
    isect_t isect;
    int k = 0;
    const int mb = sc->max_bounces;
    if (hits) for (int i = 0; i <= mb; ++i) hits[i] = PTOR_HIT_NOT_TRACED;

    ++*n_rays;
    int found = scene_intersect(sc, o, d, &isect);
    if (hits) hits[k] = found ? isect.prim : PTOR_HIT_MISS;
    ++k;
    if (!found) return v3_make(0.f, 0.f, 0.f);

    const float inv_nl = sc->n_lights ? 1.f / (float)sc->n_lights : 0.f;
    int specular_bounce = 0;
    float bsdf_pdf_prev = 0.f;
    isect_t isectp = isect;
    for (int bounces = 0;; ++bounces) {
        const ptor_sphere *prim = &sc->prims[isectp.prim];
        const ptor_material *mat = &sc->mats[prim->material_id];
        const v3 p = isectp.p;
        const v3 n = isectp.n;
        const v3 wo = d;

        if (prim->light_id > -1) {
            const ptor_light *light = &sc->lights[prim->light_id];
            v3 le = light_L(light, wo, n);
            if (bounces == 0 || specular_bounce) {
                L = v3_add(L, v3_mul(T, le));
            } else {
                /* o is still the previous vertex, wo the direction that was sampled there */
                float lp = sphere_light_pdf(sc, prim, o, wo, 0) * inv_nl;
                float w = power_heuristic(1, bsdf_pdf_prev, 1, lp);
                L = v3_add(L, v3_mul(T, v3_scale(le, w)));
            }
        }

        float rnd[4];
        ptor_rng4(sc->seed, pixel, iteration, (uint32_t)bounces, PTOR_STREAM_PATH, rnd);

        if (mat->type == 0 && sc->n_lights > 0) {
            float rl[4];
            ptor_rng4(sc->seed, pixel, iteration, (uint32_t)bounces, PTOR_STREAM_LIGHT, rl);
            uint32_t j = (uint32_t)(rl[2] * (float)sc->n_lights);
            if (j > sc->n_lights - 1) j = sc->n_lights - 1;
            const ptor_light *light = &sc->lights[j];
            v3 wl;
            float pdf_l = 0.f;   /* SampleLight leaves it unset for unknown light types (scene.h:169) */
            segment_t vt;
            v3 li = sample_light(sc, light, p, rl[0], rl[1], &wl, &pdf_l, &vt, 0);
            if (pdf_l > 0.f && !v3_is_black(li)) {
                v3 fl = material_f(mat);
                float bp = material_pdf(mat, wo, wl, n);
                if (!v3_is_black(fl) && bp > 0.f) {
                    ++*n_rays;
                    int visible;
                    if (light->type == 0) {
                        visible = !scene_intersect_p(sc, vt.o, vt.d, vt.mint, vt.maxt);
                    } else {
                        isect_t sh;
                        visible = scene_intersect(sc, vt.o, vt.d, &sh) && sh.prim == (int)light->u.prim_id;
                    }
                    if (visible) {
                        float pl = pdf_l * inv_nl;
                        float w = light->type == 0 ? 1.f : power_heuristic(1, pl, 1, bp);
                        v3 ld = v3_scale(v3_mul(fl, li), fabsf(v3_dot(wl, n)) * w / pl);
                        L = v3_add(L, v3_mul(T, ld));
                    }
                }
            }
        }

        v3 wi;
        float pdf;
        const v3 f = sample_material(mat, wo, &wi, &pdf, n, rnd[0], rnd[1]);
        if (v3_is_black(f) || pdf == 0.f) break;

        specular_bounce = mat->type != 0;                       /* pathtracer.cu:148 */
        bsdf_pdf_prev = pdf;
        T = v3_mul(T, v3_div(v3_scale(f, fabsf(v3_dot(wi, n))), pdf));
        o = p; d = wi;

        if (bounces > 3) {
            float gb = T.y > T.z ? T.y : T.z;
            float mx = T.x > gb ? T.x : gb;
            float q = .5f < mx ? .5f : mx;
            if (rnd[2] > q) break;
            T = v3_div(T, q);
        }
        if (bounces == mb) break;

        ++*n_rays;
        found = scene_intersect(sc, o, d, &isect);
        if (hits) hits[k] = found ? isect.prim : PTOR_HIT_MISS;
        ++k;
        if (!found) break;
        isectp = isect;
    }
    return L;
}

/* Component evaluation for the pin tests: `count` items, 16 floats in and 16 floats out each. */
static void nee_eval_one(const scene_ctx *sc, int op, int as_written, const float *in, float *out)
{
    v3 a, b;
    switch (op) {
    case PTOR_NEE_SAMPLE_SPHERE: a = uniform_sample_sphere(in[0], in[1]); out[0] = a.x; out[1] = a.y; out[2] = a.z; break;
    case PTOR_NEE_SAMPLE_CONE:
        a = uniform_sample_cone(in[0], in[1], in[2], v3_load(in + 3), v3_load(in + 6), v3_load(in + 9));
        out[0] = a.x; out[1] = a.y; out[2] = a.z; break;
    case PTOR_NEE_CONE_PDF: out[0] = uniform_cone_pdf(in[0]); break;
    case PTOR_NEE_POWER_HEURISTIC: out[0] = power_heuristic((int)in[0], in[1], (int)in[2], in[3]); break;
    case PTOR_NEE_COORD_SYSTEM:
        coordinate_system(v3_load(in), &a, &b);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; break;
    case PTOR_NEE_MATERIAL: {
        ptor_material m;
        m.type = (int32_t)in[0]; m.color[0] = in[1]; m.color[1] = in[2]; m.color[2] = in[3]; m.coef = 0.f;
        a = material_f(&m);
        out[0] = a.x; out[1] = a.y; out[2] = a.z;
        out[3] = material_pdf(&m, v3_load(in + 4), v3_load(in + 7), v3_load(in + 10));
        break;
    }
    case PTOR_NEE_LIGHT_L: {
        ptor_light l;
        l.type = 1; l.u.prim_id = 0; l.intensity[0] = in[0]; l.intensity[1] = in[1]; l.intensity[2] = in[2];
        a = light_L(&l, v3_load(in + 3), v3_load(in + 6));
        out[0] = a.x; out[1] = a.y; out[2] = a.z; break;
    }
    case PTOR_NEE_SEGMENT: {
        segment_t s = visibility_segment(v3_load(in), in[3], v3_load(in + 4), in[7]);
        out[0] = s.o.x; out[1] = s.o.y; out[2] = s.o.z; out[3] = s.d.x; out[4] = s.d.y; out[5] = s.d.z;
        out[6] = s.mint; out[7] = s.maxt; break;
    }
    case PTOR_NEE_SPHERE_SAMPLE_AREA:
    case PTOR_NEE_SPHERE_SAMPLE_FROM: {
        ptor_sphere s;
        s.pos[0] = in[0]; s.pos[1] = in[1]; s.pos[2] = in[2]; s.radius = in[3]; s.material_id = 0; s.light_id = -1;
        a = op == PTOR_NEE_SPHERE_SAMPLE_AREA ? sphere_sample_area(&s, in[4], in[5], &b)
                                              : sphere_sample_from(&s, v3_load(in + 4), in[7], in[8], &b, as_written);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; break;
    }
    case PTOR_NEE_LIGHT_PDF:
        out[0] = sphere_light_pdf(sc, &sc->prims[(uint32_t)in[0]], v3_load(in + 1), v3_load(in + 4), as_written); break;
    case PTOR_NEE_INTERSECT_P:
        out[0] = (float)scene_intersect_p(sc, v3_load(in), v3_load(in + 3), in[6], in[7]); break;
    case PTOR_NEE_SAMPLE_LIGHT: {
        segment_t s;
        float pdf;
        a = sample_light(sc, &sc->lights[(uint32_t)in[0]], v3_load(in + 1), in[4], in[5], &b, &pdf, &s, as_written);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; out[6] = pdf;
        out[7] = s.o.x; out[8] = s.o.y; out[9] = s.o.z; out[10] = s.d.x; out[11] = s.d.y; out[12] = s.d.z;
        out[13] = s.mint; out[14] = s.maxt; break;
    }
    default: break;
    }
}

int ptor_nee_eval(int op, int as_written,
                  const ptor_material *mats, uint32_t n_mats, const ptor_sphere *prims, uint32_t n_prims,
                  const ptor_light *lights, uint32_t n_lights, const float *in, float *out, uint32_t count)
{
    if (op < 0 || op > PTOR_NEE_SAMPLE_LIGHT || !in || !out) return -1;
    if (op >= PTOR_NEE_LIGHT_PDF && (!prims || !n_prims)) return -1;
    scene_ctx sc;
    memset(&sc, 0, sizeof sc);
    sc.mats = mats; sc.prims = prims; sc.lights = lights;
    sc.n_mats = n_mats; sc.n_prims = n_prims; sc.n_lights = n_lights;
    for (uint32_t i = 0; i < count; ++i) {
        float *o = out + (size_t)i * 16;
        for (int k = 0; k < 16; ++k) o[k] = 0.f;
        nee_eval_one(&sc, op, as_written, in + (size_t)i * 16, o);
    }
    return 0;
}
