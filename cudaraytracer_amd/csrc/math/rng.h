// Per-pixel random stream of the hot path.
//
// The reference uses cuRAND's default generator: curand_init(index + seed, 0, 0)
// then curand_uniform, range (0,1] (src/renderer/renderer.cu:357-366,
// src/math/math.h:187-188).  cuRAND is not part of the reference tree.  The
// state update below is Marsaglia's published xorwow; the seed scrambling and
// the uint32 -> float map follow cuRAND's device header as remembered, which
// cannot be checked offline, so the stream is this build's definition: the
// GPU path and the CPU checker share it, draw for draw.
#ifndef ACR_MATH_RNG_H
#define ACR_MATH_RNG_H

#include "math.h"

namespace acr
{
namespace math
{
	struct RandState
	{
		uint32_t v0, v1, v2, v3, v4, d;
	};

	// curand_init(seed, 0, 0, &state): subsequence 0 and offset 0 need no skip-ahead
	ACR_HD void randInit(RandState &s, uint64_t seed)
	{
		const uint32_t s0 = uint32_t(seed) ^ 0xaad26b49u;
		const uint32_t s1 = uint32_t(seed >> 32) ^ 0xf7dcefddu;
		const uint32_t t0 = 1099087573u * s0;
		const uint32_t t1 = 2591861531u * s1;
		s.d = 6615241u + t1 + t0;
		s.v0 = 123456789u + t0;
		s.v1 = 362436069u ^ t0;
		s.v2 = 521288629u + t1;
		s.v3 = 88675123u ^ t1;
		s.v4 = 5783321u + t0;
	}

	ACR_HD uint32_t randNext(RandState &s)
	{
		const uint32_t t = s.v0 ^ (s.v0 >> 2);
		s.v0 = s.v1;
		s.v1 = s.v2;
		s.v2 = s.v3;
		s.v3 = s.v4;
		s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
		s.d += 362437u;
		return s.v4 + s.d;
	}

	// curand_uniform: x * 2^-32 + 2^-33, two roundings, result in (0,1]
	ACR_HD float randUniform(RandState &s)
	{
		const uint32_t x = randNext(s);
		return float(x) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
	}

	// src/math/math.h:185-193.  theta runs to 2*pi, beyond the polynomial's
	// [-pi,pi] range, so the result is not unit length for theta > pi; kept.
	ACR_HD vec3 randNorm(RandState &s)
	{
		const float u = 2 * randUniform(s) - 1;
		const float theta = 2 * pi() * randUniform(s);
		const float somu = sqrtf(1 - u * u);
		return vec3(somu * fastercos(theta), somu * fastersin(theta), u);
	}

	// src/math/math.h:195-199
	ACR_HD vec3 randomHemi(const vec3 &norm, RandState &s)
	{
		const vec3 unit = randNorm(s);
		return dot(norm, unit) < 0 ? -unit : unit;
	}
} // namespace math
} // namespace acr

#endif // ACR_MATH_RNG_H
