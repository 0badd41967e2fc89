// ref_glm_kat.cpp -- known-answer generator for the GLM 0.9.6.3 primitives the
// reference's hot path uses.
//
// TEST INFRASTRUCTURE ONLY.  This driver is this repository's own code; the
// only reference files it touches are the GLM headers vendored inside the
// reference checkout, included from where they lie
// (-I/root/reference/src/libs/glm/glm, the include root src/math/math.h:10-14
// uses).  The binary goes to oracle/_ref/ and its output, a list of
// (function, input bits, output bits) records, is committed as
// tests/golden/glm_kat.json.  Both the oracle and the product's host math are
// checked bit-for-bit against those records.
//
// Build + run: see oracle/Makefile (target `ref`).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

#include "glm.hpp"
#include "gtc/matrix_transform.hpp"
#include "gtx/rotate_vector.hpp"
#include "gtc/constants.hpp"
#include "gtx/component_wise.hpp"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t next_u32()
{
	g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
	return (uint32_t)(g_state >> 32);
}
static float rnd(float lo, float hi) { return lo + (hi - lo) * (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static bool g_first = true;
static void emit(const char *fn, const std::vector<float> &in, const std::vector<float> &out)
{
	std::printf("%s{\"fn\":\"%s\",\"in\":[", g_first ? "" : ",\n", fn);
	g_first = false;
	for (size_t i = 0; i < in.size(); i++) std::printf("%s%u", i ? "," : "", bits(in[i]));
	std::printf("],\"out\":[");
	for (size_t i = 0; i < out.size(); i++) std::printf("%s%u", i ? "," : "", bits(out[i]));
	std::printf("]}");
}
static void push3(std::vector<float> &v, const glm::vec3 &a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }
static void push4(std::vector<float> &v, const glm::vec4 &a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); v.push_back(a.w); }
static void pushm4(std::vector<float> &v, const glm::mat4 &m) { for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) v.push_back(m[c][r]); }
static void pushm3(std::vector<float> &v, const glm::mat3 &m) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) v.push_back(m[c][r]); }
static glm::vec3 rv3(float s) { return glm::vec3(rnd(-s, s), rnd(-s, s), rnd(-s, s)); }
static glm::mat4 rm4_affine()
{
	// rotation * uniform scale + translation, built from plain arithmetic so the inputs are arbitrary floats
	glm::vec3 axis = glm::normalize(rv3(1.0f) + glm::vec3(0.01f, 0.02f, 0.03f));
	glm::mat4 m = glm::rotate(glm::mat4(1.0f), rnd(-3.0f, 3.0f), axis);
	float sc = rnd(0.3f, 2.5f);
	for (int c = 0; c < 3; c++) m[c] *= sc;
	m[3] = glm::vec4(rv3(8.0f), 1.0f);
	return m;
}
static glm::mat4 rm4_general()
{
	glm::mat4 m;
	for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) m[c][r] = rnd(-2.0f, 2.0f);
	return m;
}

int main()
{
	std::printf("{\"glm_version\":\"%d.%d.%d.%d\",\"cases\":[\n", GLM_VERSION_MAJOR, GLM_VERSION_MINOR, GLM_VERSION_PATCH, GLM_VERSION_REVISION);
	for (int k = 0; k < 24; k++) {
		float s = (k % 3 == 0) ? 1.0f : ((k % 3 == 1) ? 10.0f : 0.01f);
		glm::vec3 a = rv3(s), b = rv3(s);
		{ std::vector<float> in, out; push3(in, a); push3(in, b); out.push_back(glm::dot(a, b)); emit("dot3", in, out); }
		{ std::vector<float> in, out; push3(in, a); push3(in, b); push3(out, glm::cross(a, b)); emit("cross3", in, out); }
		{ std::vector<float> in, out; push3(in, a); out.push_back(glm::length(a)); emit("length3", in, out); }
		{ std::vector<float> in, out; push3(in, a); push3(out, glm::normalize(a)); emit("normalize3", in, out); }
		{ std::vector<float> in, out; glm::vec3 n = glm::normalize(b); push3(in, a); push3(in, n); push3(out, glm::reflect(a, n)); emit("reflect3", in, out); }
		{
			std::vector<float> in, out; glm::vec3 i = glm::normalize(a), n = glm::normalize(b); float eta = (k & 1) ? 1.5f : (1.0f / 1.5f);
			push3(in, i); push3(in, n); in.push_back(eta); push3(out, glm::refract(i, n, eta)); emit("refract3", in, out);
		}
		{ std::vector<float> in, out; push3(in, a); push3(in, b); push3(out, glm::min(a, b)); emit("min3", in, out); }
		{ std::vector<float> in, out; push3(in, a); push3(in, b); push3(out, glm::max(a, b)); emit("max3", in, out); }
		{ std::vector<float> in, out; push3(in, a); out.push_back(glm::compMax(a)); emit("compmax3", in, out); }
		{ std::vector<float> in, out; float d = rnd(0.1f, 9.0f); push3(in, a); in.push_back(d); push3(out, a / d); emit("div3s", in, out); }
		{ std::vector<float> in, out; push3(in, a); push3(out, 1.0f / a); emit("rdiv3", in, out); }
	}
	for (int k = 0; k < 16; k++) {
		glm::mat4 A = (k & 1) ? rm4_general() : rm4_affine();
		glm::mat4 B = (k & 2) ? rm4_general() : rm4_affine();
		glm::vec3 v = rv3(5.0f);
		glm::vec4 v4v(rv3(5.0f), rnd(-2.0f, 2.0f));
		{ std::vector<float> in, out; pushm4(in, A); push4(in, v4v); push4(out, A * v4v); emit("m4mulv", in, out); }
		{ std::vector<float> in, out; pushm4(in, A); pushm4(in, B); pushm4(out, A * B); emit("m4mul", in, out); }
		{ std::vector<float> in, out; pushm4(in, A); pushm4(out, glm::inverse(A)); emit("m4inverse", in, out); }
		{
			glm::mat3 M3(A);
			std::vector<float> in, out; pushm3(in, M3); push3(in, v); push3(out, M3 * v); emit("m3mulv", in, out);
		}
		{ glm::mat3 M3(A); std::vector<float> in, out; pushm3(in, M3); pushm3(out, glm::inverse(M3)); emit("m3inverse", in, out); }
		{
			// the normal-matrix chain of src/scene/scene.cu:366-367
			glm::mat3 N = glm::transpose(glm::inverse(glm::mat3(A)));
			glm::mat3 NI = glm::inverse(N);
			std::vector<float> in, out; pushm4(in, A); pushm3(out, N); pushm3(out, NI); emit("normal_chain", in, out);
		}
		{
			// src/math/math.h:174-178
			glm::vec4 hom = A * glm::vec4(v, 1.0f);
			glm::vec3 r = glm::vec3(hom) / hom.w;
			std::vector<float> in, out; pushm4(in, A); push3(in, v); push3(out, r); emit("translate", in, out);
		}
		{
			// src/math/math.h:180-183
			glm::mat3 M3(A);
			std::vector<float> in, out; pushm3(in, M3); push3(in, v); push3(out, glm::normalize(M3 * v)); emit("translaten", in, out);
		}
		{
			glm::vec3 axis = rv3(1.0f) + glm::vec3(0.05f, 0.0f, 0.0f);
			float ang = rnd(-1.0f, 1.0f);
			std::vector<float> in, out; push3(in, v); in.push_back(ang); push3(in, axis); push3(out, glm::rotate(v, ang, axis)); emit("rotate3", in, out);
		}
	}
	{ std::vector<float> in, out; out.push_back(glm::pi<float>()); out.push_back(2 * glm::pi<float>()); emit("pi", in, out); }
	std::printf("\n]}\n");
	return 0;
}
