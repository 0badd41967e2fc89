// Geometry value types of the hot path (reference src/geometry/geometry.h).
// Host side only: the intersection routines live in the HIP kernels
// (device/acr_kernels.hip); what remains here is what the host BIH build needs.
#ifndef ACR_GEOMETRY_H
#define ACR_GEOMETRY_H

#include <cfloat>

#include "../materials/materials.h"
#include "../math/math.h"

namespace acr
{
	struct Vertex // geometry.h:9-14
	{
		math::vec3 position;
		math::vec3 normal;
		Color3 color;
	};

	struct Ray // geometry.h:16-20
	{
		math::vec3 o;
		math::vec3 d;
	};

	struct HitInfo // geometry.h:22-27, plus the primitive id the parity tests compare
	{
		float t;
		Vertex point;
		uint32_t materialIndex;
		uint32_t primitive;
	};

	struct BoundingBox // geometry.h:29-41 (min/max alias bounds[0]/bounds[1] there)
	{
		math::vec3 min, max;
	};

	// Sum and bounds of a sequence of points (or boxes), folded strictly in the order given: the results feed the
	// BIH build, which has to come out bit-identical to the reference's, so the order of the float operations is
	// fixed -- sum += p, and min / max with the running value as first or as second operand, whichever the
	// reference writes at that place (geometry.cu:22-29, scene.cu:349-350,386-388).
	struct PointFold
	{
		math::vec3 sum = math::vec3(0, 0, 0);
		BoundingBox box = {math::vec3(FLT_MAX, FLT_MAX, FLT_MAX), math::vec3(-FLT_MAX, -FLT_MAX, -FLT_MAX)};
		size_t count = 0;

		void runningFirst(const math::vec3 &p) // min(running, p)
		{
			sum += p;
			box.min = math::min(box.min, p);
			box.max = math::max(box.max, p);
			count++;
		}
		void runningSecond(const math::vec3 &p) // min(p, running)
		{
			sum += p;
			box.min = math::min(p, box.min);
			box.max = math::max(p, box.max);
			count++;
		}
		void enclose(const BoundingBox &b) // running value first
		{
			box.min = math::min(box.min, b.min);
			box.max = math::max(box.max, b.max);
		}
		math::vec3 mean() const { return sum / float(count); }
	};

	struct Face // geometry.h:91-121
	{
		uint32_t indices[3];

		// centroid = ((0 + p0) + p1) + p2, then / 3.0f (geometry.h:95-105)
		math::vec3 getCentroid(void *verts) const
		{
			const Vertex *vs = static_cast<const Vertex *>(verts);
			math::vec3 centroid(0, 0, 0);
			for (int i = 0; i < 3; i++)
				centroid += vs[indices[i]].position;
			return centroid / 3.0f;
		}

		// geometry.h:107-121
		BoundingBox getBoundingBox(void *verts) const
		{
			const Vertex *vs = static_cast<const Vertex *>(verts);
			BoundingBox bb;
			bb.min = math::vec3(FLT_MAX, FLT_MAX, FLT_MAX);
			bb.max = math::vec3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
			for (int i = 0; i < 3; i++) {
				bb.min = math::min(bb.min, vs[indices[i]].position);
				bb.max = math::max(bb.max, vs[indices[i]].position);
			}
			return bb;
		}
	};
} // namespace acr

#endif
