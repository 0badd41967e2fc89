// Exhaustive comparison, over all 2^32 binary32 bit patterns x, of 1.0f / x as the compiler expands it (correctly
// rounded, IEEE) with a candidate shortcut: v_rcp_f32 and one Newton step.  Prints how many patterns differ and where.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_check tools/checks/rcp_check.hip && ./rcp_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcpNewton(float x)
{
	const float r0 = __builtin_amdgcn_rcpf(x);
	const float e = __builtin_fmaf(-x, r0, 1.0f);
	return __builtin_fmaf(e, r0, r0);
}

struct Report
{
	unsigned long long differ, differNormal, differResultNormal;
	uint32_t minAbs, maxAbs; // bit patterns of |x| of the smallest / largest differing normal-range argument
	uint32_t examples[8];
};

__global__ void check(Report *rep, uint32_t hi)
{
	const uint32_t bits = (hi << 24) | (blockIdx.x << 8) | threadIdx.x; // 2^24 patterns per launch
	const float x = __uint_as_float(bits);
	const float a = 1.0f / x, b = rcpNewton(x);
	const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
	const bool same = ua == ub || (a != a && b != b);
	if (!same) {
		atomicAdd(&rep->differ, 1ull);
		// does the shortcut's own result give the failure away?  (a normal number: neither zero, denormal, infinite nor NaN)
		const uint32_t eb = (ub >> 23) & 0xFFu;
		if (eb != 0 && eb != 255) atomicAdd(&rep->differResultNormal, 1ull);
		const uint32_t ax = bits & 0x7FFFFFFFu;
		// arguments whose reciprocal is a normal number: 2^-126 <= |x| <= 2^126
		if (ax >= 0x00800000u && ax <= 0x7E800000u) {
			const unsigned long long k = atomicAdd(&rep->differNormal, 1ull);
			atomicMin(&rep->minAbs, ax);
			atomicMax(&rep->maxAbs, ax);
			if (k < 8) rep->examples[k] = bits;
		}
	}
}

int main()
{
	Report *rep;
	hipMalloc(&rep, sizeof(Report));
	Report h{};
	h.minAbs = 0xFFFFFFFFu;
	hipMemcpy(rep, &h, sizeof(h), hipMemcpyHostToDevice);
	for (uint32_t hi = 0; hi < 256; hi++) hipLaunchKernelGGL(check, dim3(1u << 16), dim3(256), 0, 0, rep, hi);
	hipDeviceSynchronize();
	hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost);
	printf("differing patterns: %llu, of them with |x| in [2^-126, 2^126]: %llu (|x| bits %08x .. %08x)\n", h.differ, h.differNormal, h.minAbs, h.maxAbs);
	printf("differing patterns whose shortcut result is a normal number: %llu\n", h.differResultNormal);
	for (int k = 0; k < 8 && k < (int)h.differNormal; k++) printf("  example %08x\n", h.examples[k]);
	return 0;
}
