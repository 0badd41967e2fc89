// acr::math -- fp32 vector/matrix arithmetic shared by the C++ host side and
// the HIP device code.
//
// The reference gets these from GLM 0.9.6.3 (src/math/math.h:10-14,29).  The
// hot path's results depend on GLM's exact operation ORDER, so every function
// here spells out one IEEE binary32 operation per C++ operator in that order
// and cites the GLM line it follows (paths relative to the reference root).
// Compile with -ffp-contract=off on both g++ and hipcc: a fused multiply-add
// anywhere in this file changes results.
#ifndef ACR_MATH_MATH_H
#define ACR_MATH_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ACR_HD __host__ __device__ inline
#else
#define ACR_HD inline
#endif

namespace acr
{
namespace math
{
	struct vec2
	{
		float x, y;
		ACR_HD vec2() : x(0), y(0) {}
		ACR_HD vec2(float a, float b) : x(a), y(b) {}
	};
	ACR_HD vec2 operator*(const vec2 &a, float s) { return vec2(a.x * s, a.y * s); } // type_vec2.inl, vec * scalar
	ACR_HD vec2 operator-(const vec2 &a, const vec2 &b) { return vec2(a.x - b.x, a.y - b.y); }
	struct u32vec2 { uint32_t x, y; };
	struct u32vec3 { uint32_t x, y, z; };

	struct vec3
	{
		float x, y, z;
		ACR_HD vec3() : x(0), y(0), z(0) {}
		ACR_HD vec3(float a, float b, float c) : x(a), y(b), z(c) {}
		ACR_HD explicit vec3(float s) : x(s), y(s), z(s) {}
		// two selects in sequence (not nested): hipcc turns this shape into v_cndmask, the nested one into branches
		// members are read before the selects: a select between two member loads can be turned into one load
		// through a selected address, which pins the whole vector in memory
		ACR_HD float operator[](int i) const { const float vx = x, vy = y, vz = z; const float xy = i == 1 ? vy : vx; return i == 2 ? vz : xy; }
		ACR_HD void set(int i, float v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
	};

	struct vec4
	{
		float x, y, z, w;
		ACR_HD vec4() : x(0), y(0), z(0), w(0) {}
		ACR_HD vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
		ACR_HD vec4(const vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
		ACR_HD explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
	};

	// 1.0f / x, correctly rounded as the division is.  On the GPU the division expands into ten instructions (scaling
	// for arguments and results outside the normal range, two Newton steps, fix-ups).  The hardware's reciprocal
	// approximation followed by ONE Newton step in fused arithmetic (three instructions) returns the same bits for
	// every argument with 2^-126 <= |x| <= 2^126, and for every other argument something that is not a normal
	// number (zero, denormal, infinite, NaN).  Neither statement is an estimate: all 2^32 argument patterns have
	// been compared on gfx950, and acr_hip_selftest_reciprocal repeats that on the GPU at hand (part of the GPU
	// tests).  Hence: a normal result IS the quotient; anything else is computed again by the division.
	// Built with -DACR_NO_RCP the shortcut is the division itself (for a GPU on which the self-test fails).
#if defined(__HIPCC__)
	__device__ inline float reciprocalNewton(float x)
	{
#ifdef ACR_NO_RCP
		return 1.0f / x;
#else
		const float r = __builtin_amdgcn_rcpf(x);
		return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#endif
	}
#endif
	ACR_HD float reciprocal(float x)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const float r = reciprocalNewton(x);
		if (__builtin_isnormal(r))
			return r;
#endif
		return 1.0f / x;
	}

	// component-wise operators (type_vec3.inl / type_vec4.inl)
	ACR_HD vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
	ACR_HD vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
	ACR_HD vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
	ACR_HD vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
	ACR_HD vec3 operator*(float s, const vec3 &a) { return vec3(s * a.x, s * a.y, s * a.z); }
	ACR_HD vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }   // type_vec3.inl:707-713
	ACR_HD vec3 operator/(float s, const vec3 &a) { return vec3(s / a.x, s / a.y, s / a.z); }   // type_vec3.inl:725-731
	ACR_HD vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
	ACR_HD vec3 &operator+=(vec3 &a, const vec3 &b) { a = a + b; return a; }
	ACR_HD vec3 &operator*=(vec3 &a, float s) { a = a * s; return a; }

	ACR_HD vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
	ACR_HD vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
	ACR_HD vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
	ACR_HD vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
	ACR_HD vec4 operator/(const vec4 &a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }   // type_vec4.inl:760-766

	struct mat3
	{
		vec3 c[3]; // columns, c[col][row] like glm
		ACR_HD mat3() { c[0] = vec3(1, 0, 0); c[1] = vec3(0, 1, 0); c[2] = vec3(0, 0, 1); }
		ACR_HD float at(int col, int row) const { return c[col][row]; }
	};
	struct mat4
	{
		vec4 c[4];
		ACR_HD mat4() { c[0] = vec4(1, 0, 0, 0); c[1] = vec4(0, 1, 0, 0); c[2] = vec4(0, 0, 1, 0); c[3] = vec4(0, 0, 0, 1); }
		ACR_HD float at(int col, int row) const
		{
			const vec4 &v = c[col];
			return row == 0 ? v.x : (row == 1 ? v.y : (row == 2 ? v.z : v.w));
		}
	};

	// func_common.inl:409-435
	ACR_HD float min(float x, float y) { return x < y ? x : y; }
	ACR_HD float max(float x, float y) { return x > y ? x : y; }
	ACR_HD vec3 min(const vec3 &a, const vec3 &b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
	ACR_HD vec3 max(const vec3 &a, const vec3 &b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
	// gtx/component_wise.inl:63-69
	ACR_HD float compMax(const vec3 &v) { float r = v.x; r = max(r, v.y); r = max(r, v.z); return r; }

	// func_geometric.inl:65-72: products first, then x+y, then +z
	ACR_HD float dot(const vec3 &a, const vec3 &b) { const vec3 t = a * b; return t.x + t.y + t.z; }
	// func_geometric.inl:134-142
	ACR_HD vec3 cross(const vec3 &x, const vec3 &y)
	{
		return vec3(x.y * y.z - y.y * x.z,
		            x.z * y.x - y.z * x.x,
		            x.x * y.y - y.x * x.y);
	}
	// func_geometric.inl:95-100
	ACR_HD float length(const vec3 &v) { return sqrtf(dot(v, v)); }
	// func_geometric.inl:154-159; inversesqrt(x) = 1/sqrt(x) (func_exponential.inl:149-153)
	ACR_HD vec3 normalize(const vec3 &v) { const float inv = reciprocal(sqrtf(dot(v, v))); return v * inv; }
	// func_geometric.inl:176-179
	ACR_HD vec3 reflect(const vec3 &I, const vec3 &N) { return I - N * dot(N, I) * 2.0f; }
	// func_geometric.inl:193-200
	ACR_HD vec3 refract(const vec3 &I, const vec3 &N, float eta)
	{
		const float d = dot(N, I);
		const float k = 1.0f - eta * eta * (1.0f - d * d);
		return (eta * I - (eta * d + sqrtf(k)) * N) * float(k >= 0.0f);
	}

	// type_mat4x4.inl:617-628: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
	ACR_HD vec4 operator*(const mat4 &m, const vec4 &v)
	{
		const vec4 add0 = m.c[0] * vec4(v.x) + m.c[1] * vec4(v.y);
		const vec4 add1 = m.c[2] * vec4(v.z) + m.c[3] * vec4(v.w);
		return add0 + add1;
	}
	// type_mat3x3.inl:487-493: left to right
	ACR_HD vec3 operator*(const mat3 &m, const vec3 &v)
	{
		return vec3(m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
		            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
		            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z);
	}
	// type_mat4x4.inl:686-704
	ACR_HD mat4 operator*(const mat4 &a, const mat4 &b)
	{
		mat4 r;
		for (int j = 0; j < 4; j++)
			r.c[j] = a.c[0] * b.c[j].x + a.c[1] * b.c[j].y + a.c[2] * b.c[j].z + a.c[3] * b.c[j].w;
		return r;
	}
	ACR_HD mat3 toMat3(const mat4 &m)
	{
		mat3 r;
		for (int j = 0; j < 3; j++) r.c[j] = vec3(m.c[j].x, m.c[j].y, m.c[j].z);
		return r;
	}
	ACR_HD mat3 transpose(const mat3 &m)
	{
		mat3 r;
		r.c[0] = vec3(m.c[0].x, m.c[1].x, m.c[2].x);
		r.c[1] = vec3(m.c[0].y, m.c[1].y, m.c[2].y);
		r.c[2] = vec3(m.c[0].z, m.c[1].z, m.c[2].z);
		return r;
	}
	// type_mat3x3.inl:37-56
	inline mat3 inverse(const mat3 &m)
	{
		const float a00 = m.at(0, 0), a01 = m.at(0, 1), a02 = m.at(0, 2);
		const float a10 = m.at(1, 0), a11 = m.at(1, 1), a12 = m.at(1, 2);
		const float a20 = m.at(2, 0), a21 = m.at(2, 1), a22 = m.at(2, 2);
		const float ood = 1.0f / (+a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02) + a20 * (a01 * a12 - a11 * a02));
		mat3 r;
		r.c[0] = vec3(+(a11 * a22 - a21 * a12) * ood, -(a01 * a22 - a21 * a02) * ood, +(a01 * a12 - a11 * a02) * ood);
		r.c[1] = vec3(-(a10 * a22 - a20 * a12) * ood, +(a00 * a22 - a20 * a02) * ood, -(a00 * a12 - a10 * a02) * ood);
		r.c[2] = vec3(+(a10 * a21 - a20 * a11) * ood, -(a00 * a21 - a20 * a01) * ood, +(a00 * a11 - a10 * a01) * ood);
		return r;
	}
	// type_mat4x4.inl:37-92
	inline mat4 inverse(const mat4 &m)
	{
		const float c00 = m.at(2, 2) * m.at(3, 3) - m.at(3, 2) * m.at(2, 3);
		const float c02 = m.at(1, 2) * m.at(3, 3) - m.at(3, 2) * m.at(1, 3);
		const float c03 = m.at(1, 2) * m.at(2, 3) - m.at(2, 2) * m.at(1, 3);
		const float c04 = m.at(2, 1) * m.at(3, 3) - m.at(3, 1) * m.at(2, 3);
		const float c06 = m.at(1, 1) * m.at(3, 3) - m.at(3, 1) * m.at(1, 3);
		const float c07 = m.at(1, 1) * m.at(2, 3) - m.at(2, 1) * m.at(1, 3);
		const float c08 = m.at(2, 1) * m.at(3, 2) - m.at(3, 1) * m.at(2, 2);
		const float c10 = m.at(1, 1) * m.at(3, 2) - m.at(3, 1) * m.at(1, 2);
		const float c11 = m.at(1, 1) * m.at(2, 2) - m.at(2, 1) * m.at(1, 2);
		const float c12 = m.at(2, 0) * m.at(3, 3) - m.at(3, 0) * m.at(2, 3);
		const float c14 = m.at(1, 0) * m.at(3, 3) - m.at(3, 0) * m.at(1, 3);
		const float c15 = m.at(1, 0) * m.at(2, 3) - m.at(2, 0) * m.at(1, 3);
		const float c16 = m.at(2, 0) * m.at(3, 2) - m.at(3, 0) * m.at(2, 2);
		const float c18 = m.at(1, 0) * m.at(3, 2) - m.at(3, 0) * m.at(1, 2);
		const float c19 = m.at(1, 0) * m.at(2, 2) - m.at(2, 0) * m.at(1, 2);
		const float c20 = m.at(2, 0) * m.at(3, 1) - m.at(3, 0) * m.at(2, 1);
		const float c22 = m.at(1, 0) * m.at(3, 1) - m.at(3, 0) * m.at(1, 1);
		const float c23 = m.at(1, 0) * m.at(2, 1) - m.at(2, 0) * m.at(1, 1);

		const vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
		const vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
		const vec4 v0(m.at(1, 0), m.at(0, 0), m.at(0, 0), m.at(0, 0));
		const vec4 v1(m.at(1, 1), m.at(0, 1), m.at(0, 1), m.at(0, 1));
		const vec4 v2(m.at(1, 2), m.at(0, 2), m.at(0, 2), m.at(0, 2));
		const vec4 v3(m.at(1, 3), m.at(0, 3), m.at(0, 3), m.at(0, 3));

		const vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
		const vec4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
		const vec4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
		const vec4 i3 = v0 * f2 - v1 * f4 + v2 * f5;

		const vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
		mat4 inv;
		inv.c[0] = i0 * sa; inv.c[1] = i1 * sb; inv.c[2] = i2 * sa; inv.c[3] = i3 * sb;

		const vec4 row0(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
		const vec4 d0 = m.c[0] * row0;
		const float det = (d0.x + d0.y) + (d0.z + d0.w);
		const float ood = 1.0f / det;
		for (int j = 0; j < 4; j++) inv.c[j] = inv.c[j] * ood;
		return inv;
	}

	// gtx/rotate_vector.inl:74-82 -> gtx/transform.inl:44-50 -> gtc/matrix_transform.inl:52-85
	inline vec3 rotate(const vec3 &v, float angle, const vec3 &normal)
	{
		const float c = cosf(angle), s = sinf(angle);
		const vec3 axis = normalize(normal);
		const vec3 temp = (1.0f - c) * axis;
		float R[3][3];
		R[0][0] = c + temp.x * axis.x;
		R[0][1] = 0 + temp.x * axis.y + s * axis.z;
		R[0][2] = 0 + temp.x * axis.z - s * axis.y;
		R[1][0] = 0 + temp.y * axis.x - s * axis.z;
		R[1][1] = c + temp.y * axis.y;
		R[1][2] = 0 + temp.y * axis.z + s * axis.x;
		R[2][0] = 0 + temp.z * axis.x + s * axis.y;
		R[2][1] = 0 + temp.z * axis.y - s * axis.x;
		R[2][2] = c + temp.z * axis.z;
		const mat4 I;
		mat4 res;
		for (int j = 0; j < 3; j++)
			res.c[j] = I.c[0] * R[j][0] + I.c[1] * R[j][1] + I.c[2] * R[j][2];
		res.c[3] = I.c[3];
		return toMat3(res) * v;
	}

	// ---- the reference's own helpers (src/math/math.h) ----------------------

	// math.h:78-82 (shadows glm::epsilon)
	ACR_HD float epsilon() { return 0.001f; }

	ACR_HD uint32_t floatBits(float f) { union { float f; uint32_t i; } u; u.f = f; return u.i; }
	ACR_HD float bitsFloat(uint32_t i) { union { float f; uint32_t i; } u; u.i = i; return u.f; }

	// math.h:31-48
	ACR_HD float fastersin(float x)
	{
		const float fouroverpi = 1.2732395447351627f;
		const float fouroverpisq = 0.40528473456935109f;
		const float q = 0.77633023248007499f;
		const uint32_t xb = floatBits(x);
		const float ax = bitsFloat(xb & 0x7FFFFFFFu);
		const float p = bitsFloat(floatBits(0.22308510060189463f) | (xb & 0x80000000u));
		const float qpprox = fouroverpi * x - fouroverpisq * x * ax;
		return qpprox * (q + p * qpprox);
	}
	// math.h:50-62
	ACR_HD float fastercos(float x)
	{
		const float twooverpi = 0.63661977236758134f;
		const float p = 0.54641335845679634f;
		const float ax = bitsFloat(floatBits(x) & 0x7FFFFFFFu);
		const float qpprox = 1.0f - twooverpi * ax;
		return qpprox + p * qpprox * (1.0f - qpprox * qpprox);
	}
	// math.h:64-75
	ACR_HD float fastertanfull(float x)
	{
		const float twopi = 6.2831853071795865f;
		const float invtwopi = 0.15915494309189534f;
		const int k = int(x * invtwopi);
		const float half = (x < 0) ? -0.5f : 0.5f;
		const float xnew = x - (half + k) * twopi;
		return fastersin(xnew) / fastercos(xnew);
	}

	// math.h:126-172: two-sided Moller-Trumbore with the edge vectors already
	// formed (e1 = b - a, e2 = c - a are single subtractions, math.h:132-133,
	// so forming them once on the host gives the same bits).
	// NEWTON (device code only): the determinant's reciprocal without the division (reciprocalNewton).  The caller
	// vouches for |det| <= 2^126, the range in which that is the quotient: |det| <= |e1| |rd| |e2|, the interval-form
	// kernels call this with normalised local directions, and a scene is given to them only if |e1| |e2| < 2^120
	// holds for every triangle (checked at upload).  |det| >= eps holds past the reference's own test.
	template <bool NEWTON = false>
	ACR_HD bool myIntersectRayTriangleEdges(const vec3 &ro, const vec3 &rd, const vec3 &a, const vec3 &e1, const vec3 &e2, vec3 &baryCoords, float &t)
	{
		const vec3 P = cross(rd, e2);
		const float det = dot(e1, P);
		const float eps = epsilon();
		if (-eps < det && det < eps)
			return false;
		float invDet;
#if defined(__HIP_DEVICE_COMPILE__)
		if constexpr (NEWTON) invDet = reciprocalNewton(det);
		else invDet = 1.0f / det;
#else
		invDet = 1.0f / det;
#endif
		const vec3 T = ro - a;
		baryCoords.y = dot(T, P) * invDet;
		if (baryCoords.y < 0 || baryCoords.y > 1)
			return false;
		const vec3 Q = cross(T, e1);
		baryCoords.z = dot(rd, Q) * invDet;
		baryCoords.x = 1 - (baryCoords.y + baryCoords.z);
		if (baryCoords.z < 0 || baryCoords.x < 0)
			return false;
		t = dot(e2, Q) * invDet;
		if (t < eps)
			return false;
		return true;
	}
	ACR_HD bool myIntersectRayTriangle(const vec3 &ro, const vec3 &rd, const vec3 &a, const vec3 &b, const vec3 &c, vec3 &baryCoords, float &t)
	{
		return myIntersectRayTriangleEdges(ro, rd, a, b - a, c - a, baryCoords, t);
	}

	// math.h:174-178
	ACR_HD vec3 translate(const mat4 &m, const vec3 &v)
	{
		const vec4 hom = m * vec4(v, 1.0f);
		return vec3(hom.x, hom.y, hom.z) / hom.w;
	}
	// the same with hom.w known to be exactly 1 (bottom row (0,0,0,1)): x / 1.0f == x, the divide drops out
	ACR_HD vec3 translateAffine(const mat4 &m, const vec3 &v)
	{
		const vec4 hom = m * vec4(v, 1.0f);
		return vec3(hom.x, hom.y, hom.z);
	}
	// math.h:180-183
	ACR_HD vec3 translaten(const mat3 &m, const vec3 &v) { return normalize(m * v); }

	// math.h:201-209
	ACR_HD float rSchlick2(const vec3 &dir, const vec3 &norm, float n1, float n2)
	{
		float r0 = (n1 - n2) / (n1 + n2);
		r0 *= r0;
		const float cosI = -dot(norm, dir);
		const float x = 1 - cosI;
		return r0 + (1 - r0) * x * x * x * x * x;
	}

	// pi<float>() (gtc/constants.inl) and the 2*pi of math.h:188
	ACR_HD float pi() { return 3.14159265358979323846264338327950288f; }

} // namespace math
} // namespace acr

#endif // ACR_MATH_MATH_H
