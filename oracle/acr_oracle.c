/*
 * acr_oracle.c -- CPU oracle for the CudaRaytracer hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY: see acr_oracle.h.  PARITY UNPINNED except for the
 * GLM primitives (checked against known-answer vectors produced from the
 * reference's vendored GLM headers by oracle/ref_glm_kat.cpp).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fopenmp -fPIC -shared.
 * No -ffast-math, no -march=native: every float operation below is one IEEE
 * binary32 operation, in the order the reference source writes it.
 *
 * All "file:line" citations are relative to the reference checkout's root.
 */
#include "acr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* GLM 0.9.6.3 primitives, restated in their operation order                */
/* ======================================================================== */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c[4]; } m4; /* c[col], glm column-major */
typedef struct { v3 c[3]; } m3;

static v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
static v4 V4(float x, float y, float z, float w) { v4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static float v3get(const v3 *v, int i) { return i == 0 ? v->x : (i == 1 ? v->y : v->z); }
static void v3set(v3 *v, int i, float f) { if (i == 0) v->x = f; else if (i == 1) v->y = f; else v->z = f; }

static v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }   /* vec3 * scalar */
static v3 lscale3(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }  /* scalar * vec3 */
static v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }    /* type_vec3.inl:707-713 */
static v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static v4 sub4(v4 a, v4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static v4 mul4(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static v4 scale4(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static v4 div4s(v4 a, float s) { return V4(a.x / s, a.y / s, a.z / s, a.w / s); } /* type_vec4.inl:760-766 */

/* func_common.inl:409-435 */
static float gmin(float x, float y) { return x < y ? x : y; }
static float gmax(float x, float y) { return x > y ? x : y; }
static v3 min3(v3 a, v3 b) { return V3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static v3 max3(v3 a, v3 b) { return V3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
/* gtx/component_wise.inl:63-69 */
static float compmax3(v3 v) { float r = v.x; r = gmax(r, v.y); r = gmax(r, v.z); return r; }

/* func_geometric.inl:65-72 : tmp = x*y; tmp.x + tmp.y + tmp.z */
static float dot3(v3 a, v3 b) { v3 t = mul3(a, b); return t.x + t.y + t.z; }
/* func_geometric.inl:134-142 */
static v3 cross3(v3 x, v3 y)
{
	return V3(x.y * y.z - y.y * x.z,
	          x.z * y.x - y.z * x.x,
	          x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:95-100 */
static float length3(v3 v) { return sqrtf(dot3(v, v)); }
/* func_geometric.inl:154-159 with inversesqrt = 1/sqrt (func_exponential.inl:149-153) */
static v3 normalize3(v3 v) { float is = 1.0f / sqrtf(dot3(v, v)); return scale3(v, is); }
/* func_geometric.inl:176-179 : I - N * dot(N,I) * 2 */
static v3 reflect3(v3 I, v3 N) { return sub3(I, scale3(scale3(N, dot3(N, I)), 2.0f)); }
/* func_geometric.inl:193-200 */
static v3 refract3(v3 I, v3 N, float eta)
{
	float d = dot3(N, I);
	float k = 1.0f - eta * eta * (1.0f - d * d);
	v3 r = sub3(lscale3(eta, I), lscale3(eta * d + sqrtf(k), N));
	return scale3(r, (float)(k >= 0.0f));
}

/* type_mat4x4.inl:617-628 : (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static v4 m4mulv(const m4 *m, v4 v)
{
	v4 mul0 = mul4(m->c[0], V4(v.x, v.x, v.x, v.x));
	v4 mul1 = mul4(m->c[1], V4(v.y, v.y, v.y, v.y));
	v4 add0 = add4(mul0, mul1);
	v4 mul2 = mul4(m->c[2], V4(v.z, v.z, v.z, v.z));
	v4 mul3_ = mul4(m->c[3], V4(v.w, v.w, v.w, v.w));
	v4 add1 = add4(mul2, mul3_);
	return add4(add0, add1);
}
/* type_mat3x3.inl:487-493 : left to right */
static v3 m3mulv(const m3 *m, v3 v)
{
	return V3(m->c[0].x * v.x + m->c[1].x * v.y + m->c[2].x * v.z,
	          m->c[0].y * v.x + m->c[1].y * v.y + m->c[2].y * v.z,
	          m->c[0].z * v.x + m->c[1].z * v.y + m->c[2].z * v.z);
}
/* type_mat4x4.inl:686-704 : ((A0*b0 + A1*b1) + A2*b2) + A3*b3 per column */
static m4 m4mul(const m4 *a, const m4 *b)
{
	m4 r;
	int j;
	for (j = 0; j < 4; j++) {
		v4 bj = b->c[j];
		v4 s = add4(add4(add4(scale4(a->c[0], bj.x), scale4(a->c[1], bj.y)), scale4(a->c[2], bj.z)), scale4(a->c[3], bj.w));
		r.c[j] = s;
	}
	return r;
}
#define M4E(m, col, row) (((const float *)&(m)->c[col])[row])
/* type_mat4x4.inl:37-92 */
static m4 m4inverse(const m4 *m)
{
	float Coef00 = M4E(m,2,2) * M4E(m,3,3) - M4E(m,3,2) * M4E(m,2,3);
	float Coef02 = M4E(m,1,2) * M4E(m,3,3) - M4E(m,3,2) * M4E(m,1,3);
	float Coef03 = M4E(m,1,2) * M4E(m,2,3) - M4E(m,2,2) * M4E(m,1,3);

	float Coef04 = M4E(m,2,1) * M4E(m,3,3) - M4E(m,3,1) * M4E(m,2,3);
	float Coef06 = M4E(m,1,1) * M4E(m,3,3) - M4E(m,3,1) * M4E(m,1,3);
	float Coef07 = M4E(m,1,1) * M4E(m,2,3) - M4E(m,2,1) * M4E(m,1,3);

	float Coef08 = M4E(m,2,1) * M4E(m,3,2) - M4E(m,3,1) * M4E(m,2,2);
	float Coef10 = M4E(m,1,1) * M4E(m,3,2) - M4E(m,3,1) * M4E(m,1,2);
	float Coef11 = M4E(m,1,1) * M4E(m,2,2) - M4E(m,2,1) * M4E(m,1,2);

	float Coef12 = M4E(m,2,0) * M4E(m,3,3) - M4E(m,3,0) * M4E(m,2,3);
	float Coef14 = M4E(m,1,0) * M4E(m,3,3) - M4E(m,3,0) * M4E(m,1,3);
	float Coef15 = M4E(m,1,0) * M4E(m,2,3) - M4E(m,2,0) * M4E(m,1,3);

	float Coef16 = M4E(m,2,0) * M4E(m,3,2) - M4E(m,3,0) * M4E(m,2,2);
	float Coef18 = M4E(m,1,0) * M4E(m,3,2) - M4E(m,3,0) * M4E(m,1,2);
	float Coef19 = M4E(m,1,0) * M4E(m,2,2) - M4E(m,2,0) * M4E(m,1,2);

	float Coef20 = M4E(m,2,0) * M4E(m,3,1) - M4E(m,3,0) * M4E(m,2,1);
	float Coef22 = M4E(m,1,0) * M4E(m,3,1) - M4E(m,3,0) * M4E(m,1,1);
	float Coef23 = M4E(m,1,0) * M4E(m,2,1) - M4E(m,2,0) * M4E(m,1,1);

	v4 Fac0 = V4(Coef00, Coef00, Coef02, Coef03);
	v4 Fac1 = V4(Coef04, Coef04, Coef06, Coef07);
	v4 Fac2 = V4(Coef08, Coef08, Coef10, Coef11);
	v4 Fac3 = V4(Coef12, Coef12, Coef14, Coef15);
	v4 Fac4 = V4(Coef16, Coef16, Coef18, Coef19);
	v4 Fac5 = V4(Coef20, Coef20, Coef22, Coef23);

	v4 Vec0 = V4(M4E(m,1,0), M4E(m,0,0), M4E(m,0,0), M4E(m,0,0));
	v4 Vec1 = V4(M4E(m,1,1), M4E(m,0,1), M4E(m,0,1), M4E(m,0,1));
	v4 Vec2 = V4(M4E(m,1,2), M4E(m,0,2), M4E(m,0,2), M4E(m,0,2));
	v4 Vec3 = V4(M4E(m,1,3), M4E(m,0,3), M4E(m,0,3), M4E(m,0,3));

	v4 Inv0 = add4(sub4(mul4(Vec1, Fac0), mul4(Vec2, Fac1)), mul4(Vec3, Fac2));
	v4 Inv1 = add4(sub4(mul4(Vec0, Fac0), mul4(Vec2, Fac3)), mul4(Vec3, Fac4));
	v4 Inv2 = add4(sub4(mul4(Vec0, Fac1), mul4(Vec1, Fac3)), mul4(Vec3, Fac5));
	v4 Inv3 = add4(sub4(mul4(Vec0, Fac2), mul4(Vec1, Fac4)), mul4(Vec2, Fac5));

	v4 SignA = V4(+1, -1, +1, -1);
	v4 SignB = V4(-1, +1, -1, +1);
	m4 Inverse;
	v4 Row0, Dot0;
	float Dot1, OneOverDeterminant;
	int j;
	Inverse.c[0] = mul4(Inv0, SignA);
	Inverse.c[1] = mul4(Inv1, SignB);
	Inverse.c[2] = mul4(Inv2, SignA);
	Inverse.c[3] = mul4(Inv3, SignB);

	Row0 = V4(Inverse.c[0].x, Inverse.c[1].x, Inverse.c[2].x, Inverse.c[3].x);
	Dot0 = mul4(m->c[0], Row0);
	Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
	OneOverDeterminant = 1.0f / Dot1;
	for (j = 0; j < 4; j++) Inverse.c[j] = scale4(Inverse.c[j], OneOverDeterminant);
	return Inverse;
}
#define M3E(m, col, row) (((const float *)&(m)->c[col])[row])
#define M3W(m, col, row) (((float *)&(m)->c[col])[row])
/* type_mat3x3.inl:37-56 */
static m3 m3inverse(const m3 *m)
{
	float OneOverDeterminant = 1.0f / (
		+ M3E(m,0,0) * (M3E(m,1,1) * M3E(m,2,2) - M3E(m,2,1) * M3E(m,1,2))
		- M3E(m,1,0) * (M3E(m,0,1) * M3E(m,2,2) - M3E(m,2,1) * M3E(m,0,2))
		+ M3E(m,2,0) * (M3E(m,0,1) * M3E(m,1,2) - M3E(m,1,1) * M3E(m,0,2)));
	m3 I;
	M3W(&I,0,0) = + (M3E(m,1,1) * M3E(m,2,2) - M3E(m,2,1) * M3E(m,1,2)) * OneOverDeterminant;
	M3W(&I,1,0) = - (M3E(m,1,0) * M3E(m,2,2) - M3E(m,2,0) * M3E(m,1,2)) * OneOverDeterminant;
	M3W(&I,2,0) = + (M3E(m,1,0) * M3E(m,2,1) - M3E(m,2,0) * M3E(m,1,1)) * OneOverDeterminant;
	M3W(&I,0,1) = - (M3E(m,0,1) * M3E(m,2,2) - M3E(m,2,1) * M3E(m,0,2)) * OneOverDeterminant;
	M3W(&I,1,1) = + (M3E(m,0,0) * M3E(m,2,2) - M3E(m,2,0) * M3E(m,0,2)) * OneOverDeterminant;
	M3W(&I,2,1) = - (M3E(m,0,0) * M3E(m,2,1) - M3E(m,2,0) * M3E(m,0,1)) * OneOverDeterminant;
	M3W(&I,0,2) = + (M3E(m,0,1) * M3E(m,1,2) - M3E(m,1,1) * M3E(m,0,2)) * OneOverDeterminant;
	M3W(&I,1,2) = - (M3E(m,0,0) * M3E(m,1,2) - M3E(m,1,0) * M3E(m,0,2)) * OneOverDeterminant;
	M3W(&I,2,2) = + (M3E(m,0,0) * M3E(m,1,1) - M3E(m,1,0) * M3E(m,0,1)) * OneOverDeterminant;
	return I;
}
static m3 m3transpose(const m3 *m)
{
	m3 r;
	r.c[0] = V3(m->c[0].x, m->c[1].x, m->c[2].x);
	r.c[1] = V3(m->c[0].y, m->c[1].y, m->c[2].y);
	r.c[2] = V3(m->c[0].z, m->c[1].z, m->c[2].z);
	return r;
}
static m3 m3fromm4(const m4 *m)
{
	m3 r;
	r.c[0] = V3(m->c[0].x, m->c[0].y, m->c[0].z);
	r.c[1] = V3(m->c[1].x, m->c[1].y, m->c[1].z);
	r.c[2] = V3(m->c[2].x, m->c[2].y, m->c[2].z);
	return r;
}
/* gtx/rotate_vector.inl:74-82 -> gtx/transform.inl:44-50 -> gtc/matrix_transform.inl:52-85 */
static v3 rotate3(v3 v, float angle, v3 normal)
{
	float a = angle;
	float c = cosf(a);
	float s = sinf(a);
	v3 axis = normalize3(normal);
	v3 temp = lscale3(1.0f - c, axis);
	float R[3][3];
	m4 I, Res;
	m3 R3;
	int j;
	R[0][0] = c + temp.x * axis.x;
	R[0][1] = 0 + temp.x * axis.y + s * axis.z;
	R[0][2] = 0 + temp.x * axis.z - s * axis.y;
	R[1][0] = 0 + temp.y * axis.x - s * axis.z;
	R[1][1] = c + temp.y * axis.y;
	R[1][2] = 0 + temp.y * axis.z + s * axis.x;
	R[2][0] = 0 + temp.z * axis.x + s * axis.y;
	R[2][1] = 0 + temp.z * axis.y - s * axis.x;
	R[2][2] = c + temp.z * axis.z;
	I.c[0] = V4(1, 0, 0, 0); I.c[1] = V4(0, 1, 0, 0); I.c[2] = V4(0, 0, 1, 0); I.c[3] = V4(0, 0, 0, 1);
	for (j = 0; j < 3; j++)
		Res.c[j] = add4(add4(scale4(I.c[0], R[j][0]), scale4(I.c[1], R[j][1])), scale4(I.c[2], R[j][2]));
	Res.c[3] = I.c[3];
	R3 = m3fromm4(&Res);
	return m3mulv(&R3, v);
}

/* ======================================================================== */
/* src/math/math.h                                                          */
/* ======================================================================== */

#define ACR_EPS 0.001f /* math.h:78-82 */

typedef union { float f; uint32_t i; } fu32;

/* math.h:31-48 */
static float fastersin(float x)
{
	float fouroverpi = 1.2732395447351627f;
	float fouroverpisq = 0.40528473456935109f;
	float q = 0.77633023248007499f;
	fu32 p, vx;
	uint32_t sign;
	float qpprox;
	p.f = 0.22308510060189463f;
	vx.f = x;
	sign = vx.i & 0x80000000u;
	vx.i &= 0x7FFFFFFFu;
	qpprox = fouroverpi * x - fouroverpisq * x * vx.f;
	p.i |= sign;
	return qpprox * (q + p.f * qpprox);
}
/* math.h:50-62 */
static float fastercos(float x)
{
	float twooverpi = 0.63661977236758134f;
	float p = 0.54641335845679634f;
	fu32 vx;
	float qpprox;
	vx.f = x;
	vx.i &= 0x7FFFFFFFu;
	qpprox = 1.0f - twooverpi * vx.f;
	return qpprox + p * qpprox * (1.0f - qpprox * qpprox);
}
/* math.h:64-75 */
static float fastertanfull(float x)
{
	float twopi = 6.2831853071795865f;
	float invtwopi = 0.15915494309189534f;
	int k = (int)(x * invtwopi);
	float half = (x < 0) ? -0.5f : 0.5f;
	float xnew = x - (half + k) * twopi;
	return fastersin(xnew) / fastercos(xnew);
}

/* math.h:126-172 */
static int my_intersect_ray_triangle(v3 ro, v3 rd, v3 a, v3 b, v3 c, v3 *bary, float *t)
{
	v3 e1, e2, P, Q, T;
	float det, invDet, eps;
	e1 = sub3(b, a);
	e2 = sub3(c, a);
	P = cross3(rd, e2);
	det = dot3(e1, P);
	eps = ACR_EPS;
	if (-eps < det && det < eps)
		return 0;
	invDet = 1.0f / det;
	T = sub3(ro, a);
	bary->y = dot3(T, P) * invDet;
	if (bary->y < 0 || bary->y > 1)
		return 0;
	Q = cross3(T, e1);
	bary->z = dot3(rd, Q) * invDet;
	bary->x = 1 - (bary->y + bary->z);
	if (bary->z < 0 || bary->x < 0)
		return 0;
	*t = dot3(e2, Q) * invDet;
	if (*t < eps)
		return 0;
	return 1;
}

/* math.h:174-178 */
static v3 translate_pt(const m4 *m, v3 v)
{
	v4 hom = m4mulv(m, V4(v.x, v.y, v.z, 1.0f));
	return div3s(V3(hom.x, hom.y, hom.z), hom.w);
}
/* math.h:180-183 */
static v3 translaten(const m3 *m, v3 v) { return normalize3(m3mulv(m, v)); }

/* ---- RNG: cuRAND's default generator (XORWOW), curand_init(seed,0,0) and
 * curand_uniform as used at renderer.cu:357-366, math.h:187-188.  cuRAND is
 * not part of the reference tree; the state update is Marsaglia's published
 * xorwow, the seeding constants and the uint->float map are restated from the
 * cuRAND device header as remembered (unverifiable offline): the stream is a
 * build decision shared by this oracle and the GPU path, not a pinned fact. */
typedef struct { uint32_t v[5]; uint32_t d; } rng_t;

static void rng_init(rng_t *s, uint64_t seed)
{
	uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
	uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
	uint32_t t0 = 1099087573u * s0;
	uint32_t t1 = 2591861531u * s1;
	s->d = 6615241u + t1 + t0;
	s->v[0] = 123456789u + t0;
	s->v[1] = 362436069u ^ t0;
	s->v[2] = 521288629u + t1;
	s->v[3] = 88675123u ^ t1;
	s->v[4] = 5783321u + t0;
}
static uint32_t rng_next(rng_t *s)
{
	uint32_t t = s->v[0] ^ (s->v[0] >> 2);
	s->v[0] = s->v[1];
	s->v[1] = s->v[2];
	s->v[2] = s->v[3];
	s->v[3] = s->v[4];
	s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
	s->d += 362437u;
	return s->v[4] + s->d;
}
/* (0,1]: x * 2^-32 + 2^-33, each step rounded to binary32 */
static float rng_uniform(rng_t *s)
{
	uint32_t x = rng_next(s);
	return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* math.h:185-193 */
static v3 rand_norm(rng_t *state)
{
	float u = 2 * rng_uniform(state) - 1;
	float theta = 2 * 3.14159265358979323846264338327950288f * rng_uniform(state);
	float somu = sqrtf(1 - u * u);
	return V3(somu * fastercos(theta), somu * fastersin(theta), u);
}
/* math.h:195-199 */
static v3 random_hemi(v3 norm, rng_t *state)
{
	v3 unit = rand_norm(state);
	return dot3(norm, unit) < 0 ? neg3(unit) : unit;
}
/* math.h:201-209 */
static float r_schlick2(v3 dir, v3 norm, float n1, float n2)
{
	float r0 = (n1 - n2) / (n1 + n2);
	float cosI, x;
	r0 *= r0;
	cosI = -dot3(norm, dir);
	x = 1 - cosI;
	return r0 + (1 - r0) * x * x * x * x * x;
}

/* ======================================================================== */
/* src/geometry/geometry.h                                                   */
/* ======================================================================== */

typedef struct { v3 position, normal, color; } vertex_t;   /* geometry.h:9-14 */
typedef struct { v3 o, d; } ray_t;                          /* geometry.h:16-20 */
typedef struct {                                            /* geometry.h:22-27 (+prim) */
	float t;
	vertex_t point;
	uint32_t materialIndex;
	uint32_t prim;  /* added: global id of the triangle that set t */
} hitinfo_t;
typedef struct { v3 bounds[2]; } bbox_t;                    /* geometry.h:37-41: bounds[0]=min, bounds[1]=max */
typedef struct { v3 invD; int sign[3]; } bbargs_t;          /* geometry.h:31-35 */

typedef struct {
	orc_counters c;
	/* Path of renderer.cu:341 (bih.h:13-27), one per pixel, threaded through every Scene::intersect of the pixel */
	uint64_t path, path_length;
} stats_t;

/* Path::append as designed (bih.h:21-26; the body is commented out in the reference and tests `length` against
 * sizeof(path) = 8 where the 64 bits of `path` are meant): bits are appended from the most significant end,
 * whatever no longer fits is dropped. */
static void path_append(stats_t *st, uint64_t flag, uint64_t size)
{
	st->path_length += size;
	if (st->path_length <= 64 && flag)
		st->path |= flag << (64 - st->path_length);
}

/* geometry.h:44-76 */
static int bbox_intersect_args(const bbox_t *bb, const ray_t *r, float info_t, const bbargs_t *args, stats_t *st)
{
	const v3 *bounds = bb->bounds;
	float tmin, tmax, tymin, tymax, tzmin, tzmax;
	st->c.slab_tests++;

	tmin = (bounds[args->sign[0]].x - r->o.x) * args->invD.x;
	tmax = (bounds[1 - args->sign[0]].x - r->o.x) * args->invD.x;

	tymin = (bounds[args->sign[1]].y - r->o.y) * args->invD.y;
	tymax = (bounds[1 - args->sign[1]].y - r->o.y) * args->invD.y;

	if ((tmin > tymax) || (tymin > tmax))
		return 0;
	if (tymin > tmin)
		tmin = tymin;
	if (tymax < tmax)
		tmax = tymax;

	tzmin = (bounds[args->sign[2]].z - r->o.z) * args->invD.z;
	tzmax = (bounds[1 - args->sign[2]].z - r->o.z) * args->invD.z;

	if ((tmin > tzmax) || (tzmin > tmax))
		return 0;
	if (tzmin > tmin)
		tmin = tzmin;
	if (tzmax < tmax)
		tmax = tzmax;

	return tmin < info_t && tmax > ACR_EPS;
}
/* geometry.h:78-88 and bih.h:223-228 */
static void bbargs_make(bbargs_t *a, const ray_t *r)
{
	a->invD = V3(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
	a->sign[0] = a->invD.x < 0;
	a->sign[1] = a->invD.y < 0;
	a->sign[2] = a->invD.z < 0;
}

typedef struct { uint32_t indices[3]; uint32_t orig; } face_t;   /* geometry.h:91-93 (+orig face number) */

/* geometry.h:95-105 */
static v3 face_centroid(const face_t *f, const vertex_t *vs)
{
	v3 c = V3(0, 0, 0);
	int i;
	for (i = 0; i < 3; i++)
		c = add3(c, vs[f->indices[i]].position);
	return div3s(c, 3.0f);
}
/* geometry.h:107-121 */
static bbox_t face_bbox(const face_t *f, const vertex_t *vs)
{
	bbox_t bb;
	int i;
	bb.bounds[0] = V3(FLT_MAX, FLT_MAX, FLT_MAX);
	bb.bounds[1] = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	for (i = 0; i < 3; i++) {
		bb.bounds[0] = min3(bb.bounds[0], vs[f->indices[i]].position);
		bb.bounds[1] = max3(bb.bounds[1], vs[f->indices[i]].position);
	}
	return bb;
}
/* geometry.h:123-146 */
static int face_intersect(const face_t *f, const ray_t *r, hitinfo_t *info, const vertex_t *vertices, uint32_t prim_base, stats_t *st)
{
	const vertex_t *v0 = &vertices[f->indices[0]];
	const vertex_t *v1 = &vertices[f->indices[1]];
	const vertex_t *v2 = &vertices[f->indices[2]];
	v3 b;
	float t;
	st->c.triangle_tests++;
	if (my_intersect_ray_triangle(r->o, r->d, v0->position, v1->position, v2->position, &b, &t) && t < info->t) {
		info->t = t;
		info->point.position = add3(r->o, scale3(r->d, t));
		info->point.normal = add3(add3(lscale3(b.x, v0->normal), lscale3(b.y, v1->normal)), lscale3(b.z, v2->normal));
		info->point.color = add3(add3(lscale3(b.x, v0->color), lscale3(b.y, v1->color)), lscale3(b.z, v2->color));
		info->prim = prim_base + f->orig;
		return 1;
	}
	return 0;
}

/* ======================================================================== */
/* src/utils/bih.h                                                          */
/* ======================================================================== */

typedef struct {             /* bih.h:92-98 */
	uint32_t start, end;
	float left, right;
	uint8_t axis, isLeaf, defined;
} node_t;

typedef struct bih {
	bbox_t boundingBox;      /* bih.h:316 */
	node_t *tree;            /* bih.h:317, MAX_SIZE entries */
	uint32_t max_depth;      /* MAX_DEPTH template argument (bih.h:68) */
	uint32_t max_size;       /* bih.h:89 */
} bih_t;

/* item access for the two instantiations, BIH<Face> and BIH<Object> */
typedef struct {
	void *items;
	size_t item_size;
	bbox_t (*get_bbox)(const void *item, const void *data);
	v3 (*get_centroid)(const void *item, const void *data);
	const void *data;
} bih_items_t;

static void swap_items(char *a, char *b, size_t n)
{
	char tmp[512];
	memcpy(tmp, a, n); memcpy(a, b, n); memcpy(b, tmp, n);
}

/* bih.h:129-207 */
static int bih_sift(bih_t *bih, size_t index, const bbox_t *bbp, bih_items_t *h, size_t start, size_t end)
{
	bbox_t bb = *bbp;
	node_t *tree = bih->tree;
	float len;
	uint8_t axis;
	float pivot;
	bbox_t lBB, rBB;
	int i, j, k, left, right;

	if (index >= bih->max_size)
		return 1;

	tree[index].start = (uint32_t)start;
	tree[index].end = (uint32_t)end;
	tree[index].defined |= 1;

	if (start >= end) {
		tree[index].isLeaf = 1;
		tree[index].defined |= 4;
		return 1;
	}

	/* pick axis */
	len = -1;
	axis = 0;
	for (k = 0; k < 3; k++) {
		float axisLen = v3get(&bb.bounds[1], k) - v3get(&bb.bounds[0], k);
		if (axisLen > len) {
			len = axisLen;
			axis = (uint8_t)k;
		}
	}

	/* pick pivot */
	pivot = v3get(&bb.bounds[0], axis) + len / 2;

	lBB.bounds[0] = bb.bounds[1];
	lBB.bounds[1] = bb.bounds[0];
	rBB = lBB;

	i = (int)start;
	j = (int)end - 1;

	while (i <= j) {
		char *hi = (char *)h->items + (size_t)i * h->item_size;
		const bbox_t ibb = h->get_bbox(hi, h->data);
		const v3 cen = h->get_centroid(hi, h->data);
		float pos = v3get(&cen, axis);
		if (pos <= pivot) {
			lBB.bounds[0] = min3(lBB.bounds[0], ibb.bounds[0]);
			lBB.bounds[1] = max3(lBB.bounds[1], ibb.bounds[1]);
			i++;
		} else {
			rBB.bounds[0] = min3(rBB.bounds[0], ibb.bounds[0]);
			rBB.bounds[1] = max3(rBB.bounds[1], ibb.bounds[1]);
			if (i != j)
				swap_items(hi, (char *)h->items + (size_t)j * h->item_size, h->item_size);
			j--;
		}
	}

	tree[index].left = v3get(&lBB.bounds[1], axis);
	tree[index].right = v3get(&rBB.bounds[0], axis);
	tree[index].axis = axis;
	tree[index].defined |= 2;

	left = bih_sift(bih, 2 * index + 1, &lBB, h, start, (size_t)i);
	right = bih_sift(bih, 2 * index + 2, &rBB, h, (size_t)i, end);
	tree[index].isLeaf = (uint8_t)(left || right);
	tree[index].defined |= 4;
	return 0;
}

#define ORC_MAX_STACK 40 /* >= MAX_DEPTH+2 (bih.h:214-215) for every depth this oracle accepts */

typedef int (*item_intersect_fn)(const void *items, uint32_t i, const ray_t *r, hitinfo_t *info, const void *data, stats_t *st);

/* bih.h:209-299.  record_path: the traversal appends to the pixel's Path (bih.h:268-274,293-297); only the
 * tree over objects is called with a Path in the reference (scene.cu:206), the mesh trees are not. */
static int bih_tree_intersect(const bih_t *bih, const void *objs, item_intersect_fn item_intersect,
                              const ray_t *r, hitinfo_t *info, const void *data, stats_t *st, int record_path)
{
	int intersected = 0;
	uint64_t bitsAppended = 0;
	int treeIdx[ORC_MAX_STACK];
	bbox_t treeBB[ORC_MAX_STACK];
	int depth = 0;
	bbargs_t bbArgs;
	const node_t *tree = bih->tree;

	treeIdx[0] = 0;
	treeBB[0] = bih->boundingBox;

	bbargs_make(&bbArgs, r);

	while (depth >= 0) {
		/* traverse tree */
		while (depth >= 0) {
			int i = treeIdx[depth];
			const bbox_t bb = treeBB[depth];
			const node_t *n = &tree[i];
			bbox_t lBB, rBB;
			int hitL, hitR, lIdx, rIdx;

			if (n->isLeaf) break;
			st->c.node_visits++;

			lBB = bb;
			rBB = bb;
			v3set(&lBB.bounds[1], n->axis, n->left);
			v3set(&rBB.bounds[0], n->axis, n->right);

			hitL = bbox_intersect_args(&lBB, r, info->t, &bbArgs, st);
			hitR = bbox_intersect_args(&rBB, r, info->t, &bbArgs, st);

			lIdx = 2 * i + 1;
			rIdx = 2 * i + 2;

			if (hitR) {
				treeIdx[depth] = rIdx;
				treeBB[depth] = rBB;
				depth++;
			}
			if (hitL) {
				treeIdx[depth] = lIdx;
				treeBB[depth] = lBB;
				depth++;
			}
			depth--;

			if (record_path) {
				/* left got more than right (bih.h:269-270) */
				const int lBigger = (tree[lIdx].end - tree[lIdx].start) > (tree[rIdx].end - tree[rIdx].start);
				/* 00 neither hit, 11 both, 10 the bigger child, 01 the smaller one (bih.h:272-274) */
				path_append(st, ((uint64_t)hitL << lBigger) | ((uint64_t)hitR << !lBigger), 2);
				bitsAppended += 2;
			}
		}

		if (depth >= 0) {
			int i = treeIdx[depth];
			const node_t *n = &tree[i];
			/* intersect objects */
			if (n->isLeaf) {
				uint32_t k;
				for (k = n->start; k < n->end; k++)
					intersected |= item_intersect(objs, k, r, info, data, st);
				depth--;
			}
		}
	}
	if (record_path) { /* bih.h:293-297: every traversal takes 2 * MAX_SIZE bits of the Path's length */
		const uint64_t total = 2ull * bih->max_size;
		if (total > bitsAppended) path_append(st, 0, total - bitsAppended);
	}
	return intersected;
}

/* ======================================================================== */
/* src/geometry/mesh.h, geometry.cu                                          */
/* ======================================================================== */

typedef struct mesh {
	v3 centroid;
	bbox_t boundingBox;
	vertex_t *vertices; uint32_t nverts;
	bih_t faces;        face_t *face_items; uint32_t nfaces;
	uint32_t materialIndex;
	uint32_t prim_base;
	int near_first;     /* not in the reference: see mesh_bih_intersect_near */
} mesh_t;

static bbox_t face_bbox_cb(const void *item, const void *data) { return face_bbox((const face_t *)item, (const vertex_t *)data); }
static v3 face_centroid_cb(const void *item, const void *data) { return face_centroid((const face_t *)item, (const vertex_t *)data); }

static void bih_init(bih_t *b, uint32_t depth, const bbox_t *bb)
{
	b->max_depth = depth;
	b->max_size = (uint32_t)((1ull << depth) - 1);
	b->boundingBox = *bb;
	b->tree = (node_t *)calloc(b->max_size ? b->max_size : 1, sizeof(node_t));
}

/* geometry.cu:8-48 */
static void mesh_build(mesh_t *m, const orc_import_scene *imp, const orc_import_mesh *am, uint32_t depth)
{
	uint32_t i, j;
	v3 sumVertices = V3(0, 0, 0);
	v3 minBound = V3(FLT_MAX, FLT_MAX, FLT_MAX);
	v3 maxBound = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	bih_items_t items;

	m->materialIndex = am->material_index;
	m->nverts = am->vertex_count;
	m->vertices = (vertex_t *)malloc(sizeof(vertex_t) * (am->vertex_count ? am->vertex_count : 1));
	for (i = 0; i < am->vertex_count; i++) {
		const float *p = imp->positions + 3 * (size_t)(am->vertex_begin + i);
		const float *n = imp->normals + 3 * (size_t)(am->vertex_begin + i);
		const float *c = imp->colors ? imp->colors + 3 * (size_t)(am->vertex_begin + i) : NULL;
		vertex_t *v = &m->vertices[i];
		v->position = V3(p[0], p[1], p[2]);
		v->normal = V3(n[0], n[1], n[2]);
		v->color = c ? V3(c[0], c[1], c[2]) : V3(1.0f, 1.0f, 1.0f);
		sumVertices = add3(sumVertices, v->position);
		minBound = min3(minBound, v->position);
		maxBound = max3(maxBound, v->position);
	}
	m->boundingBox.bounds[0] = minBound;
	m->boundingBox.bounds[1] = maxBound;
	m->centroid = div3s(sumVertices, (float)am->vertex_count);

	m->nfaces = am->face_count;
	m->prim_base = am->face_begin;
	m->face_items = (face_t *)malloc(sizeof(face_t) * (am->face_count ? am->face_count : 1));
	for (i = 0; i < am->face_count; i++) {
		for (j = 0; j < 3; j++)
			m->face_items[i].indices[j] = imp->indices[3 * (size_t)(am->face_begin + i) + j];
		m->face_items[i].orig = i;
	}
	bih_init(&m->faces, depth, &m->boundingBox);
	items.items = m->face_items;
	items.item_size = sizeof(face_t);
	items.get_bbox = face_bbox_cb;
	items.get_centroid = face_centroid_cb;
	items.data = m->vertices;
	bih_sift(&m->faces, 0, &m->faces.boundingBox, &items, 0, m->nfaces);
}

typedef struct { const vertex_t *vertices; uint32_t prim_base; } face_ctx_t;

static int face_item_intersect(const void *items, uint32_t i, const ray_t *r, hitinfo_t *info, const void *data, stats_t *st)
{
	const face_ctx_t *fc = (const face_ctx_t *)data;
	return face_intersect(&((const face_t *)items)[i], r, info, fc->vertices, fc->prim_base, st);
}


/* ---- NOT in the reference: the same tree walked near side first ----------
 * The GPU path walks deep mesh trees visiting the child on the ray's near side first and dropping boxes that
 * start behind the current hit.  This restates that order on the CPU for two purposes: counting the node
 * visits and triangle tests of that order (the work the GPU path's roofline is priced on), and checking on the
 * CPU that it returns what bih_tree_intersect returns.  The rules are the GPU path's:
 *   - a box is kept while its tmin lies below the current hit distance times ORC_NEAR_SLACK;
 *   - an exact tie goes to the item with the lower position in the item array (the one bih_tree_intersect finds
 *     first), but never against a hit of an earlier mesh of the same object;
 *   - if another hit X has t_W < t_X <= tmin(leaf of the winner W), the order could have mattered (the
 *     reference may have dropped W's leaf): the mesh is walked again in the reference's order and counted in
 *     order_doubts. */
#define ORC_NEAR_SLACK (1.0f + 1.0f / 32768.0f)

/* geometry.h:44-76 with the entry distance returned */
static int bbox_intersect_tmin(const bbox_t *bb, const ray_t *r, float info_t, const bbargs_t *args, stats_t *st, float *tmin_out)
{
	const v3 *bounds = bb->bounds;
	float tmin, tmax, tymin, tymax, tzmin, tzmax;
	st->c.slab_tests++;
	tmin = (bounds[args->sign[0]].x - r->o.x) * args->invD.x;
	tmax = (bounds[1 - args->sign[0]].x - r->o.x) * args->invD.x;
	tymin = (bounds[args->sign[1]].y - r->o.y) * args->invD.y;
	tymax = (bounds[1 - args->sign[1]].y - r->o.y) * args->invD.y;
	if ((tmin > tymax) || (tymin > tmax))
		return 0;
	if (tymin > tmin)
		tmin = tymin;
	if (tymax < tmax)
		tmax = tymax;
	tzmin = (bounds[args->sign[2]].z - r->o.z) * args->invD.z;
	tzmax = (bounds[1 - args->sign[2]].z - r->o.z) * args->invD.z;
	if ((tmin > tzmax) || (tzmin > tmax))
		return 0;
	if (tzmin > tmin)
		tmin = tzmin;
	if (tzmax < tmax)
		tmax = tzmax;
	*tmin_out = tmin;
	return tmin < info_t && tmax > ACR_EPS;
}

static int mesh_bih_intersect_near(const mesh_t *m, const ray_t *r, hitinfo_t *info, stats_t *st)
{
	const bih_t *bih = &m->faces;
	const node_t *tree = bih->tree;
	const hitinfo_t entry = *info;
	int intersected = 0;
	int treeIdx[ORC_MAX_STACK];
	bbox_t treeBB[ORC_MAX_STACK];
	float treeTmin[ORC_MAX_STACK];
	int depth = 0;
	bbargs_t bbArgs;
	float second = FLT_MAX, leafTmin = 0.0f;
	uint32_t bestItem = 0;

	bbargs_make(&bbArgs, r);
	treeIdx[0] = 0;
	treeBB[0] = bih->boundingBox;
	if (!bbox_intersect_tmin(&treeBB[0], r, info->t * ORC_NEAR_SLACK, &bbArgs, st, &treeTmin[0]))
		return 0; /* cannot happen after Mesh::intersect's own test of the same box */

	while (depth >= 0) {
		const int i = treeIdx[depth];
		const bbox_t bb = treeBB[depth];
		const float tminB = treeTmin[depth];
		const node_t *n = &tree[i];
		depth--;
		if (!(tminB < info->t * ORC_NEAR_SLACK))
			continue; /* a waiting box that now starts behind the hit */
		if (!n->isLeaf) {
			bbox_t cBB[2];
			float cT[2];
			int hit[2], nearSide, farSide;
			st->c.node_visits++;
			cBB[0] = bb;
			cBB[1] = bb;
			v3set(&cBB[0].bounds[1], n->axis, n->left);
			v3set(&cBB[1].bounds[0], n->axis, n->right);
			hit[0] = bbox_intersect_tmin(&cBB[0], r, info->t * ORC_NEAR_SLACK, &bbArgs, st, &cT[0]);
			hit[1] = bbox_intersect_tmin(&cBB[1], r, info->t * ORC_NEAR_SLACK, &bbArgs, st, &cT[1]);
			nearSide = bbArgs.sign[n->axis] ? 1 : 0; /* the left child is the near one for a positive direction */
			farSide = 1 - nearSide;
			if (hit[farSide]) {
				depth++;
				treeIdx[depth] = 2 * i + 1 + farSide;
				treeBB[depth] = cBB[farSide];
				treeTmin[depth] = cT[farSide];
			}
			if (hit[nearSide]) {
				depth++;
				treeIdx[depth] = 2 * i + 1 + nearSide;
				treeBB[depth] = cBB[nearSide];
				treeTmin[depth] = cT[nearSide];
			}
		} else {
			uint32_t k;
			for (k = n->start; k < n->end; k++) {
				const face_t *f = &m->face_items[k];
				const vertex_t *v0 = &m->vertices[f->indices[0]];
				const vertex_t *v1 = &m->vertices[f->indices[1]];
				const vertex_t *v2 = &m->vertices[f->indices[2]];
				v3 b;
				float t;
				int take, tie;
				st->c.triangle_tests++;
				if (!my_intersect_ray_triangle(r->o, r->d, v0->position, v1->position, v2->position, &b, &t))
					continue;
				take = t < info->t;
				tie = t == info->t;
				if (take) second = info->t;
				else if (!tie && t < second) second = t;
				take = take || (tie && intersected && k < bestItem);
				if (take) {
					leafTmin = tminB;
					bestItem = k;
					info->t = t;
					info->point.position = add3(r->o, scale3(r->d, t));
					info->point.normal = add3(add3(lscale3(b.x, v0->normal), lscale3(b.y, v1->normal)), lscale3(b.z, v2->normal));
					info->point.color = add3(add3(lscale3(b.x, v0->color), lscale3(b.y, v1->color)), lscale3(b.z, v2->color));
					info->prim = m->prim_base + f->orig;
					intersected = 1;
				}
			}
		}
	}
	if (second < FLT_MAX && second <= leafTmin) {
		face_ctx_t fc;
		st->c.order_doubts++;
		*info = entry;
		fc.vertices = m->vertices;
		fc.prim_base = m->prim_base;
		return bih_tree_intersect(&m->faces, m->face_items, face_item_intersect, r, info, &fc, st, 0);
	}
	return intersected;
}

/* geometry.cu:52-61 */
static int mesh_intersect(const mesh_t *m, const ray_t *r, hitinfo_t *info, stats_t *st)
{
	bbargs_t args;
	face_ctx_t fc;
	st->c.mesh_visits++;
	bbargs_make(&args, r); /* geometry.h:78-88 */
	fc.vertices = m->vertices;
	fc.prim_base = m->prim_base;
	if (bbox_intersect_args(&m->boundingBox, r, info->t, &args, st)
	    && (st->c.mesh_box_hits++, 1)
	    && (m->near_first ? mesh_bih_intersect_near(m, r, info, st)
	                      : bih_tree_intersect(&m->faces, m->face_items, face_item_intersect, r, info, &fc, st, 0))) {
		info->materialIndex = m->materialIndex;
		return 1;
	}
	return 0;
}

/* ======================================================================== */
/* src/scene/scene.h, scene.cu                                               */
/* ======================================================================== */

typedef struct object {          /* scene.h:16-50 */
	int index, parentIndex;
	v3 centroid;
	bbox_t boundingBox;
	int *meshes; uint32_t nmeshes;
	m4 globalTransform, localTransform;
	m3 globalNormalTransform;
	m4 globalInverseTransform;
	m3 globalInverseNormalTransform;
} object_t;

typedef struct light {           /* scene.h:52-80 */
	int type;
	v3 ambient, diffuse, specular, position, direction;
	float attConstant, attLinear, attQuadratic, innerConeAngle, outerConeAngle;
} light_t;

typedef struct material { v3 ambient, diffuse, specular; float refractiveIndex; } material_t; /* materials.h:12-18 */

struct orc_scene {
	bih_t objects;           /* scene.h:117 */
	object_t *object_items; uint32_t nobjects;      /* filtered objects, permuted by sift */
	material_t *materials; uint32_t nmaterials;
	mesh_t *meshes; uint32_t nmeshes;
	light_t *lights; uint32_t nlights;
	bbox_t boundingBox;
	int rootIndex;
	orc_camera camera;
	uint32_t ntriangles;
};

/* scene.cu:332-357 */
static bbox_t object_transform_bbox(const object_t *o, bbox_t bb)
{
	bbox_t bounds;
	const float *coords = (const float *)&bb;
	int i;
	bounds.bounds[1] = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	bounds.bounds[0] = V3(FLT_MAX, FLT_MAX, FLT_MAX);
	for (i = 0; i < 8; i++) {
		int xi = 3 * (i >> 2);
		int yi = 3 * ((i >> 1) & 1) + 1;
		int zi = 3 * (i & 1) + 2;
		v3 localVertex = V3(coords[xi], coords[yi], coords[zi]);
		v3 globalVertex = translate_pt(&o->globalTransform, localVertex);
		bounds.bounds[0] = min3(globalVertex, bounds.bounds[0]);
		bounds.bounds[1] = max3(globalVertex, bounds.bounds[1]);
	}
	return bounds;
}

/* scene.cu:359-405 */
static void object_build(object_t *o, const orc_import_scene *imp, const orc_import_node *node, int index,
                         const object_t *parent, const mesh_t *hMeshes)
{
	uint32_t i;
	m3 g3, inv3;
	o->index = index;
	o->parentIndex = parent ? parent->index : -1;
	memcpy(&o->localTransform, node->transform, sizeof(m4));
	o->globalTransform = parent ? m4mul(&parent->globalTransform, &o->localTransform) : o->localTransform;
	o->globalInverseTransform = m4inverse(&o->globalTransform);
	g3 = m3fromm4(&o->globalTransform);
	inv3 = m3inverse(&g3);
	o->globalNormalTransform = m3transpose(&inv3);
	o->globalInverseNormalTransform = m3inverse(&o->globalNormalTransform);

	o->nmeshes = node->mesh_count;
	o->meshes = (int *)malloc(sizeof(int) * (node->mesh_count ? node->mesh_count : 1));
	for (i = 0; i < node->mesh_count; i++)
		o->meshes[i] = (int)imp->node_mesh_ids[node->mesh_begin + i];
	o->centroid = V3(0, 0, 0);
	o->boundingBox.bounds[0] = V3(0, 0, 0);
	o->boundingBox.bounds[1] = V3(0, 0, 0);

	if (o->nmeshes > 0) {
		v3 minBound = V3(FLT_MAX, FLT_MAX, FLT_MAX);
		v3 maxBound = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
		v3 sumCentroids = V3(0, 0, 0);
		bbox_t localBoundingBox;
		v3 avgCentroid;
		for (i = 0; i < o->nmeshes; i++) {
			const mesh_t *m = &hMeshes[o->meshes[i]];
			sumCentroids = add3(sumCentroids, m->centroid);
			minBound = min3(minBound, m->boundingBox.bounds[0]);
			maxBound = max3(maxBound, m->boundingBox.bounds[1]);
		}
		localBoundingBox.bounds[0] = minBound;
		localBoundingBox.bounds[1] = maxBound;
		o->boundingBox = object_transform_bbox(o, localBoundingBox);
		avgCentroid = div3s(sumCentroids, (float)o->nmeshes);
		o->centroid = translate_pt(&o->globalTransform, avgCentroid);
	}
}

/* scene.cu:407-440 */
static int object_intersect(const object_t *o, const ray_t *r, hitinfo_t *info, const mesh_t *meshMap, stats_t *st)
{
	ray_t lr;
	int intersected = 0;
	hitinfo_t tmpInfo;
	uint32_t i;
	st->c.object_visits++;
	lr.o = translate_pt(&o->globalInverseTransform, r->o);
	lr.d = translaten(&o->globalInverseNormalTransform, r->d);

	memset(&tmpInfo, 0, sizeof(tmpInfo));
	tmpInfo.t = FLT_MAX;

	for (i = 0; i < o->nmeshes; i++)
		intersected |= mesh_intersect(&meshMap[o->meshes[i]], &lr, &tmpInfo, st);

	if (intersected) {
		float t;
		st->c.object_hits++;
		tmpInfo.point.position = translate_pt(&o->globalTransform, tmpInfo.point.position);
		tmpInfo.point.normal = translaten(&o->globalNormalTransform, tmpInfo.point.normal);
		t = length3(sub3(tmpInfo.point.position, r->o));
		if (t < info->t) {
			*info = tmpInfo;
			info->t = t;
		}
	}
	return intersected;
}

static int object_item_intersect(const void *items, uint32_t i, const ray_t *r, hitinfo_t *info, const void *data, stats_t *st)
{
	return object_intersect(&((const object_t *)items)[i], r, info, (const mesh_t *)data, st);
}
static bbox_t object_bbox_cb(const void *item, const void *data) { (void)data; return ((const object_t *)item)->boundingBox; }
static v3 object_centroid_cb(const void *item, const void *data) { (void)data; return ((const object_t *)item)->centroid; }

/* scene.cu:204-207 */
static int scene_intersect(const orc_scene *s, const ray_t *r, hitinfo_t *info, stats_t *st)
{
	return bih_tree_intersect(&s->objects, s->object_items, object_item_intersect, r, info, s->meshes, st, 1);
}

/* scene.cu:209-231 */
static v3 point_light_accum(const orc_scene *s, const light_t *l, v3 pos, v3 norm, rng_t *state, stats_t *st)
{
	v3 dir = sub3(add3(l->position, lscale3(0.5f, rand_norm(state))), pos);
	float t = length3(dir);
	float cosTheta;
	v3 c;
	ray_t r;
	hitinfo_t info;

	dir = normalize3(dir);

	cosTheta = gmax(dot3(dir, norm), 0.0f);
	c = lscale3(cosTheta / (l->attConstant + (l->attLinear + l->attQuadratic * t) * t), l->diffuse);
	if (length3(c) < ACR_EPS) return c;

	r.o = pos;
	r.d = dir;

	memset(&info, 0, sizeof(info));
	info.t = t;
	st->c.rays_shadow++;
	if (scene_intersect(s, &r, &info, st) && info.t + ACR_EPS < t)
		return V3(0, 0, 0);
	return c;
}

/* scene.cu:234-271 */
static v3 spot_light_accum(const orc_scene *s, const light_t *l, v3 pos, v3 norm, stats_t *st)
{
	v3 dir = sub3(l->position, pos);
	float t = length3(dir);
	float cosTheta, cosLight, lightTheta, inner, outer, falloff_distance, val, falloff;
	v3 c;
	ray_t r;
	hitinfo_t info;

	dir = normalize3(dir);

	cosTheta = gmax(dot3(dir, norm), 0.0f);
	cosLight = dot3(neg3(dir), normalize3(l->direction));
	lightTheta = acosf(cosLight);
	inner = l->outerConeAngle - 1.57079632679f;
	outer = l->innerConeAngle;
	falloff_distance = outer - inner;

	if (fabsf(lightTheta) >= outer / 2.0f)
		return V3(0, 0, 0);

	c = lscale3(cosTheta / (l->attConstant + (l->attLinear + l->attQuadratic * t) * t), l->diffuse);
	if (length3(c) < ACR_EPS) return c;

	val = (outer / 2.0f - fabsf(lightTheta));
	falloff = val / (falloff_distance / 2.0f);

	if ((double)fabsf(lightTheta) >= (double)inner / 2.0)
		c = scale3(c, falloff);

	r.o = pos;
	r.d = dir;

	memset(&info, 0, sizeof(info));
	info.t = t;
	st->c.rays_shadow++;
	if (scene_intersect(s, &r, &info, st) && info.t + ACR_EPS < t)
		return V3(0, 0, 0);
	return c;
}

/* scene.cu:273-292 */
static v3 light_point(const orc_scene *s, v3 pos, v3 norm, rng_t *state, stats_t *st)
{
	v3 light = V3(0, 0, 0);
	uint32_t i;
	for (i = 0; i < s->nlights; i++) {
		const light_t *l = &s->lights[i];
		switch (l->type) {
		case 1: /* DIRECTIONAL */
		case 2: /* SPOT */
			light = add3(light, spot_light_accum(s, l, pos, norm, st));
			break;
		case 0: /* POINT */
			light = add3(light, point_light_accum(s, l, pos, norm, state, st));
			break;
		}
	}
	return light;
}

/* ---- scene construction (scene.cu:54-202, 294-330; materials.cpp:5-17) --- */

typedef struct {
	const orc_import_scene *imp;
	orc_scene *s;
	object_t *objs; uint32_t nobjs;
	uint32_t *child_begin, *child_list; /* children of node n: child_list[child_begin[n] .. child_begin[n+1]) */
} load_ctx_t;

static int name_eq(const char *a, const char *b) { return strncmp(a, b, 64) == 0; }

/* scene.cu:155-202 */
static int load_object(load_ctx_t *ctx, uint32_t node_idx, const object_t *parent)
{
	const orc_import_scene *imp = ctx->imp;
	const orc_import_node *node = &imp->nodes[node_idx];
	object_t tmp;
	int i;
	uint32_t k, li;

	object_build(&tmp, imp, node, (int)ctx->nobjs, parent, ctx->s->meshes);
	ctx->objs[ctx->nobjs++] = tmp;
	i = tmp.index;

	/* lights are matched by node name (scene.cu:174-183); the map keeps the first light of a name (scene.cu:117) */
	for (li = 0; li < imp->num_lights; li++) {
		if (name_eq(imp->lights[li].name, node->name)) {
			uint32_t lj, first = li;
			for (lj = 0; lj < li; lj++)
				if (name_eq(imp->lights[lj].name, node->name)) { first = lj; break; }
			if (first == li) {
				light_t *l = &ctx->s->lights[li];
				l->position = translate_pt(&ctx->objs[i].globalTransform, l->position);
				l->direction = translaten(&tmp.globalNormalTransform, l->direction);
			}
			break;
		}
	}
	/* scene.cu:185-191 */
	if (name_eq(node->name, imp->camera.name)) {
		orc_camera *c = &ctx->s->camera;
		v3 p = translate_pt(&tmp.globalTransform, V3(c->position[0], c->position[1], c->position[2]));
		v3 f = translaten(&tmp.globalNormalTransform, V3(c->forward[0], c->forward[1], c->forward[2]));
		v3 u = translaten(&tmp.globalNormalTransform, V3(c->up[0], c->up[1], c->up[2]));
		c->position[0] = p.x; c->position[1] = p.y; c->position[2] = p.z;
		c->forward[0] = f.x; c->forward[1] = f.y; c->forward[2] = f.z;
		c->up[0] = u.x; c->up[1] = u.y; c->up[2] = u.z;
	}

	for (k = ctx->child_begin[node_idx]; k < ctx->child_begin[node_idx + 1]; k++)
		load_object(ctx, ctx->child_list[k], &tmp);
	return i;
}

orc_scene *orc_scene_create(const orc_import_scene *imp, uint32_t scene_bih_depth, uint32_t mesh_bih_depth)
{
	orc_scene *s;
	load_ctx_t ctx;
	uint32_t i, n, nf;
	bih_items_t items;

	if (!imp || scene_bih_depth < 1 || scene_bih_depth > ORC_MAX_STACK - 2 || mesh_bih_depth < 1 || mesh_bih_depth > ORC_MAX_STACK - 2)
		return NULL;
	s = (orc_scene *)calloc(1, sizeof(orc_scene));

	/* scene.cu:294-301 */
	s->camera.aspect = imp->camera.aspect;
	s->camera.hfov = imp->camera.hfov;
	memcpy(s->camera.position, imp->camera.position, 12);
	memcpy(s->camera.up, imp->camera.up, 12);
	memcpy(s->camera.forward, imp->camera.look_at, 12);

	/* scene.cu:303-330 */
	s->nlights = imp->num_lights;
	s->lights = (light_t *)calloc(s->nlights ? s->nlights : 1, sizeof(light_t));
	for (i = 0; i < s->nlights; i++) {
		const orc_import_light *a = &imp->lights[i];
		light_t *l = &s->lights[i];
		l->attConstant = a->att_constant; l->attLinear = a->att_linear; l->attQuadratic = a->att_quadratic;
		l->innerConeAngle = a->inner_cone; l->outerConeAngle = a->outer_cone;
		l->position = V3(a->position[0], a->position[1], a->position[2]);
		l->direction = V3(a->direction[0], a->direction[1], a->direction[2]);
		l->ambient = V3(a->ambient[0], a->ambient[1], a->ambient[2]);
		l->diffuse = V3(a->diffuse[0], a->diffuse[1], a->diffuse[2]);
		l->specular = V3(a->specular[0], a->specular[1], a->specular[2]);
		l->type = (a->type == 1 || a->type == 2) ? a->type : 0;
	}

	/* materials.cpp:5-17 */
	s->nmaterials = imp->num_materials;
	s->materials = (material_t *)calloc(s->nmaterials ? s->nmaterials : 1, sizeof(material_t));
	for (i = 0; i < s->nmaterials; i++) {
		const orc_import_material *a = &imp->materials[i];
		s->materials[i].diffuse = V3(a->diffuse[0], a->diffuse[1], a->diffuse[2]);
		s->materials[i].ambient = V3(a->ambient[0], a->ambient[1], a->ambient[2]);
		s->materials[i].specular = V3(a->specular[0], a->specular[1], a->specular[2]);
		s->materials[i].refractiveIndex = a->refracti;
	}

	/* scene.cu:89-92 */
	s->nmeshes = imp->num_meshes;
	s->meshes = (mesh_t *)calloc(s->nmeshes ? s->nmeshes : 1, sizeof(mesh_t));
	for (i = 0; i < s->nmeshes; i++) {
		mesh_build(&s->meshes[i], imp, &imp->meshes[i], mesh_bih_depth);
		s->ntriangles += s->meshes[i].nfaces;
	}

	/* scene.cu:122-152 */
	ctx.imp = imp;
	ctx.s = s;
	ctx.objs = (object_t *)calloc(imp->num_nodes ? imp->num_nodes : 1, sizeof(object_t));
	ctx.nobjs = 0;
	ctx.child_begin = (uint32_t *)calloc(imp->num_nodes + 2, sizeof(uint32_t));
	ctx.child_list = (uint32_t *)calloc(imp->num_nodes + 1, sizeof(uint32_t));
	for (i = 0; i < imp->num_nodes; i++)
		if (imp->nodes[i].parent >= 0) ctx.child_begin[imp->nodes[i].parent + 1]++;
	for (i = 0; i < imp->num_nodes; i++) ctx.child_begin[i + 1] += ctx.child_begin[i];
	{
		uint32_t *fill = (uint32_t *)calloc(imp->num_nodes + 1, sizeof(uint32_t));
		for (i = 0; i < imp->num_nodes; i++) {
			int p = imp->nodes[i].parent;
			if (p >= 0) ctx.child_list[ctx.child_begin[p] + fill[p]++] = i;
		}
		free(fill);
	}
	s->rootIndex = imp->num_nodes ? load_object(&ctx, 0, NULL) : -1;

	s->boundingBox.bounds[1] = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	s->boundingBox.bounds[0] = V3(FLT_MAX, FLT_MAX, FLT_MAX);

	n = 0;
	for (i = 0; i < ctx.nobjs; i++) if (ctx.objs[i].nmeshes > 0) n++;
	s->nobjects = n;
	s->object_items = (object_t *)calloc(n ? n : 1, sizeof(object_t));
	nf = 0;
	for (i = 0; i < ctx.nobjs; i++) {
		if (ctx.objs[i].nmeshes > 0) s->object_items[nf++] = ctx.objs[i];
		else free(ctx.objs[i].meshes);
	}
	for (i = 0; i < s->nobjects; i++) {
		s->boundingBox.bounds[1] = max3(s->boundingBox.bounds[1], s->object_items[i].boundingBox.bounds[1]);
		s->boundingBox.bounds[0] = min3(s->boundingBox.bounds[0], s->object_items[i].boundingBox.bounds[0]);
	}
	bih_init(&s->objects, scene_bih_depth, &s->boundingBox);
	items.items = s->object_items;
	items.item_size = sizeof(object_t);
	items.get_bbox = object_bbox_cb;
	items.get_centroid = object_centroid_cb;
	items.data = NULL;
	bih_sift(&s->objects, 0, &s->objects.boundingBox, &items, 0, s->nobjects);

	free(ctx.objs);
	free(ctx.child_begin);
	free(ctx.child_list);
	return s;
}

void orc_scene_destroy(orc_scene *s)
{
	uint32_t i;
	if (!s) return;
	for (i = 0; i < s->nmeshes; i++) {
		free(s->meshes[i].vertices);
		free(s->meshes[i].face_items);
		free(s->meshes[i].faces.tree);
	}
	for (i = 0; i < s->nobjects; i++) free(s->object_items[i].meshes);
	free(s->objects.tree);
	free(s->object_items);
	free(s->meshes);
	free(s->materials);
	free(s->lights);
	free(s);
}

/* ======================================================================== */
/* src/renderer/renderer.cu                                                  */
/* ======================================================================== */

/* renderer.cu:184-206 */
static v3 get_pixel_dir(const orc_camera *camera, float ni, float nj)
{
	v3 dir = V3(camera->forward[0], camera->forward[1], camera->forward[2]);
	v3 up = V3(camera->up[0], camera->up[1], camera->up[2]);
	float AR = camera->aspect;
	v3 cR = cross3(dir, up);
	v3 cU = cross3(cR, dir);
	float dist = fastertanfull(camera->hfov / 2.0f);
	return normalize3(add3(dir, lscale3(dist, add3(lscale3(nj, cU), lscale3(AR * ni, cR)))));
}

typedef struct { int hit; hitinfo_t info; } primary_rec_t;

/* renderer.cu:232-324.  N is the template argument; the state is advanced in
 * place, which is what the reference's by-value copies amount to because no
 * caller reads its own copy after the recursive call.  nSource is left
 * uninitialised by the reference's translucent branch (renderer.cu:251,
 * 286-311); this build defines it as `source`. */
static v4 ray_color(int N, const ray_t *r, v3 source, const orc_scene *scene, rng_t *state, stats_t *st, primary_rec_t *rec)
{
	hitinfo_t info;
	if (N <= 0)
		return V4(0, 0, 0, 1);
	memset(&info, 0, sizeof(info));
	info.t = FLT_MAX;
	if (scene_intersect(scene, r, &info, st)) {
		const material_t *mat = &scene->materials[info.materialIndex];
		v3 c, nSource;
		ray_t nr;
		v4 rest;
		if (rec) { rec->hit = 1; rec->info = info; }

		/* direct illum */
		c = add3(mat->ambient, mul3(mat->diffuse, light_point(scene, info.point.position, info.point.normal, state, st)));

		nSource = source;
		nr.o = info.point.position;
		nr.d = V3(0, 0, 0);

		if (mat->refractiveIndex == 0) {
			v3 cd = mul3(source, mat->diffuse);
			v3 cs = mul3(source, mat->specular);
			float Pd = compmax3(cd) / compmax3(source);
			float Ps = compmax3(cs) / compmax3(source);
			float P = rng_uniform(state);
			if (P < Pd) {
				nSource = scale3(cd, 1 / Pd);
				nr.d = random_hemi(info.point.normal, state);
			} else if (P < Pd + Ps) {
				nSource = scale3(cs, 1 / Ps);
				nr.d = reflect3(r->d, info.point.normal);
			} else {
				return V4(c.x, c.y, c.z, 1);
			}
		} else {
			float n1, n2, R;
			v3 norm = info.point.normal;
			if (dot3(r->d, norm) < 0) {
				n1 = mat->refractiveIndex;
				n2 = 1;
			} else {
				n1 = 1;
				n2 = mat->refractiveIndex;
				norm = scale3(norm, -1.0f);
			}
			R = r_schlick2(r->d, norm, n1, n2);
			if (rng_uniform(state) >= R)
				nr.d = refract3(r->d, norm, n1 / n2);
			else
				nr.d = reflect3(r->d, norm);
		}
		if (N - 1 > 0) st->c.rays_bounce++;
		rest = ray_color(N - 1, &nr, nSource, scene, state, st, NULL);
		return add4(V4(c.x, c.y, c.z, 1), rest);
	}
	if (rec) rec->hit = 0;
	return V4(0, 0, 0, 1);
}

/* renderer.cu:326-388, one pixel slot */
static void scatter_trace_pixel(const orc_scene *scene, const orc_camera *cam, const orc_frame *f, int oldIndex,
                                float *screen, const int32_t *pixel_map, const orc_aov *aov, stats_t *st)
{
	const int width = (int)f->width, height = (int)f->height;
	int index, x, y;
	rng_t state;
	float dx, dy, i, j;
	ray_t r;
	v4 contribution, *px;
	primary_rec_t rec;

	if (f->frames == 0 || !pixel_map)
		index = oldIndex;
	else
		index = pixel_map[oldIndex];

	x = index % width;
	y = index / width;

	rng_init(&state, (uint64_t)(int64_t)index + f->seed);

	dx = 1.0f / width;
	dy = 1.0f / height;

	i = 2.0f * ((float)x + rng_uniform(&state)) * dx - 1.0f;
	j = 2.0f * ((float)y + rng_uniform(&state)) * dy - 1.0f;

	r.o = V3(cam->position[0], cam->position[1], cam->position[2]);
	r.d = get_pixel_dir(cam, i, -j);

	rec.hit = 0;
	st->c.rays_primary++;
	st->path = 0;          /* Path path; (renderer.cu:341) */
	st->path_length = 0;
	contribution = ray_color(f->max_bounces, &r, V3(1, 1, 1), scene, &state, st, &rec);
	if (rec.hit) st->c.primary_hits++;

	px = (v4 *)(screen + 4 * (size_t)index);
	if (f->frames > 0) {
		*px = scale4(*px, (float)f->frames / (float)(f->frames + 1));
		*px = add4(*px, div4s(contribution, (float)(f->frames + 1)));
	} else {
		*px = contribution;
	}

	if (aov) {
		size_t k = (size_t)index;
		if (aov->path_key) aov->path_key[k] = st->path;
		if (aov->prim) aov->prim[k] = rec.hit ? rec.info.prim : 0xFFFFFFFFu;
		if (aov->t) aov->t[k] = rec.hit ? rec.info.t : FLT_MAX;
		if (aov->material) aov->material[k] = rec.hit ? rec.info.materialIndex : 0xFFFFFFFFu;
		if (aov->position) {
			aov->position[3 * k + 0] = rec.hit ? rec.info.point.position.x : 0.0f;
			aov->position[3 * k + 1] = rec.hit ? rec.info.point.position.y : 0.0f;
			aov->position[3 * k + 2] = rec.hit ? rec.info.point.position.z : 0.0f;
		}
		if (aov->normal) {
			aov->normal[3 * k + 0] = rec.hit ? rec.info.point.normal.x : 0.0f;
			aov->normal[3 * k + 1] = rec.hit ? rec.info.point.normal.y : 0.0f;
			aov->normal[3 * k + 2] = rec.hit ? rec.info.point.normal.z : 0.0f;
		}
	}
	(void)height;
}

static void counters_add(orc_counters *a, const orc_counters *b)
{
	a->rays_primary += b->rays_primary; a->rays_shadow += b->rays_shadow; a->rays_bounce += b->rays_bounce;
	a->primary_hits += b->primary_hits; a->node_visits += b->node_visits; a->slab_tests += b->slab_tests;
	a->object_visits += b->object_visits; a->mesh_visits += b->mesh_visits; a->triangle_tests += b->triangle_tests;
	a->object_hits += b->object_hits; a->mesh_box_hits += b->mesh_box_hits;
	a->order_doubts += b->order_doubts;
}

int orc_render(const orc_scene *s, const orc_camera *cam_in, const orc_frame *f,
               float *screen, const int32_t *pixel_map, const orc_aov *aov, orc_counters *counters)
{
	orc_camera cam;
	orc_counters total;
	int row_begin, row_end, row;
	if (!s || !cam_in || !f || !screen || f->width == 0 || f->height == 0) return 1;
	cam = *cam_in;
	cam.aspect = (float)f->width / (float)f->height; /* renderer.cu:179 */
	memset(&total, 0, sizeof(total));
	row_begin = (int)f->row_begin;
	row_end = (int)(f->row_end ? f->row_end : f->height);
	if (row_end > (int)f->height) row_end = (int)f->height;
#ifdef _OPENMP
	if (f->num_threads > 0) omp_set_num_threads(f->num_threads);
#endif
#pragma omp parallel
	{
		stats_t st;
		memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 1)
		for (row = row_begin; row < row_end; row++) {
			int col;
			for (col = 0; col < (int)f->width; col++)
				scatter_trace_pixel(s, &cam, f, col + (int)f->width * row, screen, pixel_map, aov, &st);
		}
#pragma omp critical
		counters_add(&total, &st.c);
	}
	if (counters) *counters = total;
	return 0;
}

void orc_trace_rays(const orc_scene *s, uint32_t n, const float *origins, const float *dirs,
                    const float *tmax, uint8_t *hit, float *t, uint32_t *prim, float *position, float *normal)
{
	uint32_t i;
	stats_t st;
	memset(&st, 0, sizeof(st));
	for (i = 0; i < n; i++) {
		ray_t r;
		hitinfo_t info;
		int h;
		r.o = V3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
		r.d = V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
		memset(&info, 0, sizeof(info));
		info.t = tmax ? tmax[i] : FLT_MAX;
		info.prim = 0xFFFFFFFFu;
		h = scene_intersect(s, &r, &info, &st);
		if (hit) hit[i] = (uint8_t)h;
		if (t) t[i] = info.t;
		if (prim) prim[i] = info.prim;
		if (position) { position[3 * i] = info.point.position.x; position[3 * i + 1] = info.point.position.y; position[3 * i + 2] = info.point.position.z; }
		if (normal) { normal[3 * i] = info.point.normal.x; normal[3 * i + 1] = info.point.normal.y; normal[3 * i + 2] = info.point.normal.z; }
	}
}

/* ======================================================================== */
/* inspection + unit entry points                                            */
/* ======================================================================== */

void orc_scene_set_mesh_order(orc_scene *s, int near_first)
{
	uint32_t i;
	for (i = 0; i < s->nmeshes; i++) s->meshes[i].near_first = near_first != 0;
}
void orc_scene_camera(const orc_scene *s, orc_camera *out) { *out = s->camera; }
uint32_t orc_scene_num_objects(const orc_scene *s) { return s->nobjects; }
uint32_t orc_scene_num_meshes(const orc_scene *s) { return s->nmeshes; }
uint32_t orc_scene_num_lights(const orc_scene *s) { return s->nlights; }
uint32_t orc_scene_num_triangles(const orc_scene *s) { return s->ntriangles; }
void orc_scene_bounds(const orc_scene *s, float mn[3], float mx[3])
{
	memcpy(mn, &s->boundingBox.bounds[0], 12);
	memcpy(mx, &s->boundingBox.bounds[1], 12);
}
static void export_nodes(const bih_t *b, orc_bih_node *nodes)
{
	uint32_t i;
	for (i = 0; i < b->max_size; i++) {
		const node_t *n = &b->tree[i];
		nodes[i].start = n->start; nodes[i].end = n->end;
		nodes[i].left = n->left; nodes[i].right = n->right;
		nodes[i].axis = n->axis; nodes[i].is_leaf = n->isLeaf; nodes[i].defined = n->defined; nodes[i].pad = 0;
	}
}
void orc_scene_bih(const orc_scene *s, orc_bih_node *nodes, int32_t *object_order)
{
	uint32_t i;
	if (nodes) export_nodes(&s->objects, nodes);
	if (object_order) for (i = 0; i < s->nobjects; i++) object_order[i] = s->object_items[i].index;
}
void orc_scene_object(const orc_scene *s, uint32_t i, float out[59])
{
	const object_t *o = &s->object_items[i];
	memcpy(out, &o->globalTransform, 64);
	memcpy(out + 16, &o->globalInverseTransform, 64);
	memcpy(out + 32, &o->globalNormalTransform, 36);
	memcpy(out + 41, &o->globalInverseNormalTransform, 36);
	memcpy(out + 50, &o->centroid, 12);
	memcpy(out + 53, &o->boundingBox, 24);
}
void orc_mesh_bih(const orc_scene *s, uint32_t mesh, orc_bih_node *nodes, uint32_t *face_order)
{
	const mesh_t *m = &s->meshes[mesh];
	uint32_t i;
	if (nodes) export_nodes(&m->faces, nodes);
	if (face_order) for (i = 0; i < m->nfaces; i++) face_order[i] = m->face_items[i].orig;
}
void orc_mesh_info(const orc_scene *s, uint32_t mesh, float centroid[3], float mn[3], float mx[3], uint32_t *num_faces, uint32_t *material)
{
	const mesh_t *m = &s->meshes[mesh];
	if (centroid) memcpy(centroid, &m->centroid, 12);
	if (mn) memcpy(mn, &m->boundingBox.bounds[0], 12);
	if (mx) memcpy(mx, &m->boundingBox.bounds[1], 12);
	if (num_faces) *num_faces = m->nfaces;
	if (material) *material = m->materialIndex;
}
void orc_scene_light(const orc_scene *s, uint32_t i, int32_t *type, float out[14])
{
	const light_t *l = &s->lights[i];
	if (type) *type = l->type;
	memcpy(out, &l->position, 12);
	memcpy(out + 3, &l->direction, 12);
	memcpy(out + 6, &l->diffuse, 12);
	out[9] = l->attConstant; out[10] = l->attLinear; out[11] = l->attQuadratic;
	out[12] = l->innerConeAngle; out[13] = l->outerConeAngle;
}

float orc_fastersin(float x) { return fastersin(x); }
float orc_fastercos(float x) { return fastercos(x); }
float orc_fastertanfull(float x) { return fastertanfull(x); }
int orc_ray_triangle(const float ro[3], const float rd[3], const float a[3], const float b[3], const float c[3], float bary[3], float *t)
{
	v3 bc = V3(0, 0, 0);
	float tt = 0;
	int h = my_intersect_ray_triangle(V3(ro[0], ro[1], ro[2]), V3(rd[0], rd[1], rd[2]), V3(a[0], a[1], a[2]),
	                                  V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]), &bc, &tt);
	bary[0] = bc.x; bary[1] = bc.y; bary[2] = bc.z;
	*t = tt;
	return h;
}
int orc_slab(const float mn[3], const float mx[3], const float ro[3], const float rd[3], float tcur)
{
	bbox_t bb;
	ray_t r;
	bbargs_t a;
	stats_t st;
	memset(&st, 0, sizeof(st));
	bb.bounds[0] = V3(mn[0], mn[1], mn[2]);
	bb.bounds[1] = V3(mx[0], mx[1], mx[2]);
	r.o = V3(ro[0], ro[1], ro[2]);
	r.d = V3(rd[0], rd[1], rd[2]);
	bbargs_make(&a, &r);
	return bbox_intersect_args(&bb, &r, tcur, &a, &st);
}
void orc_rng_uniforms(uint64_t seed, uint32_t n, float *out)
{
	rng_t s;
	uint32_t i;
	rng_init(&s, seed);
	for (i = 0; i < n; i++) out[i] = rng_uniform(&s);
}
void orc_translate(const float m[16], const float v[3], float out[3])
{
	m4 M; v3 r;
	memcpy(&M, m, 64);
	r = translate_pt(&M, V3(v[0], v[1], v[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_translaten(const float m[9], const float v[3], float out[3])
{
	m3 M; v3 r;
	memcpy(&M, m, 36);
	r = translaten(&M, V3(v[0], v[1], v[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_mat4_inverse(const float m[16], float out[16]) { m4 M, I; memcpy(&M, m, 64); I = m4inverse(&M); memcpy(out, &I, 64); }
void orc_mat3_inverse(const float m[9], float out[9]) { m3 M, I; memcpy(&M, m, 36); I = m3inverse(&M); memcpy(out, &I, 36); }
void orc_mat4_mul(const float a[16], const float b[16], float out[16]) { m4 A, B, C; memcpy(&A, a, 64); memcpy(&B, b, 64); C = m4mul(&A, &B); memcpy(out, &C, 64); }
void orc_normalize(const float v[3], float out[3]) { v3 r = normalize3(V3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_reflect(const float i[3], const float n[3], float out[3]) { v3 r = reflect3(V3(i[0], i[1], i[2]), V3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_refract(const float i[3], const float n[3], float eta, float out[3]) { v3 r = refract3(V3(i[0], i[1], i[2]), V3(n[0], n[1], n[2]), eta); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_rotate(const float v[3], float angle, const float axis[3], float out[3]) { v3 r = rotate3(V3(v[0], v[1], v[2]), angle, V3(axis[0], axis[1], axis[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }

/* renderer.cu:145-166 */
void orc_move_camera(orc_camera *cam, const float pos[2], const float dir[2])
{
	v3 forward = V3(cam->forward[0], cam->forward[1], cam->forward[2]);
	v3 up = V3(cam->up[0], cam->up[1], cam->up[2]);
	v3 position = V3(cam->position[0], cam->position[1], cam->position[2]);
	v3 right, delta;
	forward = rotate3(forward, dir[0], up);
	right = cross3(forward, up);
	forward = rotate3(forward, dir[1], right);
	up = rotate3(up, dir[1], right);
	delta = add3(scale3(forward, pos[1]), scale3(right, pos[0]));
	position = add3(position, delta);
	cam->forward[0] = forward.x; cam->forward[1] = forward.y; cam->forward[2] = forward.z;
	cam->up[0] = up.x; cam->up[1] = up.y; cam->up[2] = up.z;
	cam->position[0] = position.x; cam->position[1] = position.y; cam->position[2] = position.z;
}

/* Dispatcher for the known-answer records of tests/golden/glm_kat.json:
 * `in`/`out` are the flat float lists of one record.  Returns the number of
 * outputs written, or -1 for an unknown function name. */
int orc_kat(const char *fn, const float *in, float *out)
{
	v3 a = V3(in[0], in[1], in[2]);
	if (!strcmp(fn, "dot3")) { out[0] = dot3(a, V3(in[3], in[4], in[5])); return 1; }
	if (!strcmp(fn, "cross3")) { v3 r = cross3(a, V3(in[3], in[4], in[5])); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "length3")) { out[0] = length3(a); return 1; }
	if (!strcmp(fn, "normalize3")) { v3 r = normalize3(a); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "reflect3")) { v3 r = reflect3(a, V3(in[3], in[4], in[5])); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "refract3")) { v3 r = refract3(a, V3(in[3], in[4], in[5]), in[6]); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "min3")) { v3 r = min3(a, V3(in[3], in[4], in[5])); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "max3")) { v3 r = max3(a, V3(in[3], in[4], in[5])); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "compmax3")) { out[0] = compmax3(a); return 1; }
	if (!strcmp(fn, "div3s")) { v3 r = div3s(a, in[3]); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "rdiv3")) { out[0] = 1.0f / a.x; out[1] = 1.0f / a.y; out[2] = 1.0f / a.z; return 3; }
	if (!strcmp(fn, "m4mulv")) { m4 M; v4 r; memcpy(&M, in, 64); r = m4mulv(&M, V4(in[16], in[17], in[18], in[19])); memcpy(out, &r, 16); return 4; }
	if (!strcmp(fn, "m4mul")) { m4 A, B, C; memcpy(&A, in, 64); memcpy(&B, in + 16, 64); C = m4mul(&A, &B); memcpy(out, &C, 64); return 16; }
	if (!strcmp(fn, "m4inverse")) { orc_mat4_inverse(in, out); return 16; }
	if (!strcmp(fn, "m3mulv")) { m3 M; v3 r; memcpy(&M, in, 36); r = m3mulv(&M, V3(in[9], in[10], in[11])); memcpy(out, &r, 12); return 3; }
	if (!strcmp(fn, "m3inverse")) { orc_mat3_inverse(in, out); return 9; }
	if (!strcmp(fn, "normal_chain")) {
		m4 A; m3 g3, inv3, N, NI;
		memcpy(&A, in, 64);
		g3 = m3fromm4(&A); inv3 = m3inverse(&g3); N = m3transpose(&inv3); NI = m3inverse(&N);
		memcpy(out, &N, 36); memcpy(out + 9, &NI, 36);
		return 18;
	}
	if (!strcmp(fn, "translate")) { orc_translate(in, in + 16, out); return 3; }
	if (!strcmp(fn, "translaten")) { orc_translaten(in, in + 9, out); return 3; }
	if (!strcmp(fn, "rotate3")) { orc_rotate(in, in[3], in + 4, out); return 3; }
	if (!strcmp(fn, "pi")) { out[0] = 3.14159265358979323846264338327950288f; out[1] = 2 * 3.14159265358979323846264338327950288f; return 2; }
	return -1;
}

/* PathAggregate (bih.h:29-66), whose bodies the reference has commented out, as those bodies read once their two
 * slips are set right: set()'s loop counts `i--` from 0 (bih.h:48, it would stop after one counter) and the
 * conversion shifts once more after the last counter (bih.h:60-63, which drops counter 0 and reverses the order).
 * What the design needs is that an aggregate of one path converts back to that path: counter i belongs to bit i.
 * One aggregate per PIXEL (the reference indexes them by thread slot, renderer.cu:380,386, so that after a
 * re-sort a slot would keep voting with another pixel's history). */
void orc_path_votes(const uint64_t *path_keys, int32_t *votes, uint32_t n, int reset)
{
	uint32_t px, i;
	for (px = 0; px < n; px++) {
		uint64_t bits = path_keys[px];
		for (i = 0; i < 64; i++) {
			const int32_t vote = (bits & 1u) ? -1 : 1;          /* bih.h:39,49 */
			int32_t *w = &votes[(size_t)i * n + px];
			*w = reset ? vote : *w + vote;
			bits >>= 1;
		}
	}
}

void orc_vote_keys(const int32_t *votes, uint64_t *keys, uint32_t n)
{
	uint32_t px, i;
	for (px = 0; px < n; px++) {
		uint64_t path = 0;
		for (i = 0; i < 64; i++)
			if (votes[(size_t)i * n + px] < 0) path |= (uint64_t)1 << i; /* bih.h:61 */
		keys[px] = path;
	}
}
