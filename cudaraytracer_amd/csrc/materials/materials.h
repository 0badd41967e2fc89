// acr::Material (reference src/materials/materials.h:7-25, materials.cpp:5-17)
#ifndef ACR_MATERIALS_H
#define ACR_MATERIALS_H

#include "../../../include/acr_host.h"
#include "../math/math.h"

namespace acr
{
	typedef math::vec3 Color3;
	typedef math::vec4 Color4;

	class Material
	{
	public:
		Material() : refractiveIndex(0) {}
		// the reference reads AI_MATKEY_COLOR_DIFFUSE/AMBIENT/SPECULAR and
		// AI_MATKEY_REFRACTI from an aiMaterial (materials.cpp:7-16); a missing
		// REFRACTI key leaves its field uninitialised there, the import layer
		// answers 0 (opaque) instead.
		explicit Material(const acr_import_material &m)
			: ambient(m.ambient[0], m.ambient[1], m.ambient[2]), diffuse(m.diffuse[0], m.diffuse[1], m.diffuse[2]),
			  specular(m.specular[0], m.specular[1], m.specular[2]), refractiveIndex(m.refracti)
		{
		}
		Color3 ambient, diffuse, specular;
		float refractiveIndex;
	};
} // namespace acr

#endif
