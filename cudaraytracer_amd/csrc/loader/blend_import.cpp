// .blend -> import-level scene, standing in for the reference's
//   importer.ReadFile(path, aiProcess_Triangulate | aiProcess_JoinIdenticalVertices | aiProcess_SortByPType)
// (src/scene/scene.cu:30-35).  assimp is an empty submodule in the reference
// tree, so the conventions below follow assimp 3.x's Blender importer as far
// as they are known and are otherwise this loader's own, stated decisions:
//   * one root node "<BlenderRoot>" (identity); one node per object of the
//     scene's base list, in list order, children under their parent; a node's
//     transform is inverse(parent obmat) * obmat (obmat for top-level objects);
//   * a mesh object yields one mesh per material slot in use (ascending slot);
//     every polygon corner is a vertex (position MVert.co, normal
//     MVert.no / 32767), triangles fan out from a polygon's first corner,
//     then corners equal in position and normal are merged, first come first
//     numbered;
//   * materials are numbered in order of first use; diffuse (r,g,b), specular
//     (specr,specg,specb), ambient (ambr,ambg,ambb), no refraction index
//     (answered as 0 = opaque); meshes without a material share a default one
//     (diffuse and specular 0.6 grey, ambient 0);
//   * lamps: LOCAL -> point, SUN -> directional, SPOT -> spot; colours are
//     (r,g,b) * energy; attenuation constant 0, linear 1, quadratic 0; cone
//     angles 2*pi (spot: spotsize); local position (0,0,0), direction (0,0,-1)
//     for sun and spot;
//   * camera: the first camera object; position (0,0,0), up (0,1,0), look-at
//     (0,0,-1), horizontal FOV pi/4, aspect 0 (the renderer overrides it).
#include <cmath>
#include <map>

#include "../math/math.h"
#include "blend_reader.h"
#include "import_scene.h"

namespace acr
{
	namespace
	{
		struct Obj
		{
			uint64_t ptr = 0, parent = 0, data = 0;
			int type = 0;
			std::string name;
			math::mat4 obmat;
			std::vector<size_t> children;
		};

		void copyName(char (&dst)[64], const std::string &s)
		{
			std::memset(dst, 0, 64);
			std::memcpy(dst, s.data(), s.size() < 63 ? s.size() : 63);
		}

		struct Converter
		{
			const blend::File &file;
			ImportScene &out;
			std::vector<Obj> objs;
			std::map<uint64_t, uint32_t> materialIndex; // Material block address -> index in out.materials
			int defaultMaterial = -1;
			bool haveCamera = false;
			std::string error;

			Converter(const blend::File &f, ImportScene &o) : file(f), out(o) {}

			uint32_t materialFor(uint64_t matPtr)
			{
				const blend::Block *mb = file.blockAt(matPtr);
				const blend::Struct *ms = mb ? file.structOf(*mb) : nullptr;
				if (!mb || !ms || ms->name != "Material") {
					if (defaultMaterial < 0) {
						acr_import_material m{};
						for (int k = 0; k < 3; k++) { m.diffuse[k] = 0.6f; m.specular[k] = 0.6f; m.ambient[k] = 0.0f; }
						m.refracti = 0.0f;
						defaultMaterial = int(out.materials.size());
						out.materials.push_back(m);
					}
					return uint32_t(defaultMaterial);
				}
				auto it = materialIndex.find(matPtr);
				if (it != materialIndex.end()) return it->second;
				const uint8_t *p = file.payload(*mb);
				acr_import_material m{};
				m.diffuse[0] = file.read<float>(p, *ms, "r"); m.diffuse[1] = file.read<float>(p, *ms, "g"); m.diffuse[2] = file.read<float>(p, *ms, "b");
				m.specular[0] = file.read<float>(p, *ms, "specr"); m.specular[1] = file.read<float>(p, *ms, "specg"); m.specular[2] = file.read<float>(p, *ms, "specb");
				m.ambient[0] = file.read<float>(p, *ms, "ambr"); m.ambient[1] = file.read<float>(p, *ms, "ambg"); m.ambient[2] = file.read<float>(p, *ms, "ambb");
				m.refracti = 0.0f;
				const uint32_t idx = uint32_t(out.materials.size());
				out.materials.push_back(m);
				materialIndex.emplace(matPtr, idx);
				return idx;
			}

			// one mesh object -> meshes appended to out; returns their indices
			bool convertMesh(const Obj &o, std::vector<uint32_t> &meshIds)
			{
				const blend::Block *mb = file.blockAt(o.data);
				const blend::Struct *ms = mb ? file.structOf(*mb) : nullptr;
				if (!mb || !ms || ms->name != "Mesh") return true; // no geometry
				const uint8_t *mp = file.payload(*mb);
				const int totvert = file.read<int32_t>(mp, *ms, "totvert");
				const int totpoly = file.read<int32_t>(mp, *ms, "totpoly");
				const int totloop = file.read<int32_t>(mp, *ms, "totloop");
				const int totcol = file.read<int16_t>(mp, *ms, "totcol");
				const blend::Block *vb = file.blockAt(file.readPointer(mp, *ms, "mvert"));
				const blend::Block *pb = file.blockAt(file.readPointer(mp, *ms, "mpoly"));
				const blend::Block *lb = file.blockAt(file.readPointer(mp, *ms, "mloop"));
				const blend::Block *matb = file.blockAt(file.readPointer(mp, *ms, "mat"));
				if (totpoly <= 0 || totvert <= 0) return true;
				const blend::Struct *vs = vb ? file.structOf(*vb) : nullptr;
				const blend::Struct *ps = pb ? file.structOf(*pb) : nullptr;
				const blend::Struct *ls = lb ? file.structOf(*lb) : nullptr;
				if (!vs || !ps || !ls || vs->name != "MVert" || ps->name != "MPoly" || ls->name != "MLoop" ||
				    vb->count < uint32_t(totvert) || pb->count < uint32_t(totpoly) || lb->count < uint32_t(totloop) ||
				    file.payloadAvailable(*vb) < vs->size * size_t(totvert) || file.payloadAvailable(*pb) < ps->size * size_t(totpoly) ||
				    file.payloadAvailable(*lb) < ls->size * size_t(totloop)) {
					error = "mesh '" + o.name + "': vertex/polygon/loop blocks missing or short";
					return false;
				}
				const uint8_t *vp = file.payload(*vb), *pp = file.payload(*pb), *lp = file.payload(*lb);

				std::vector<int> slots; // material slots in use, ascending
				for (int p = 0; p < totpoly; p++) {
					const int nr = file.read<int16_t>(pp + ps->size * size_t(p), *ps, "mat_nr");
					bool seen = false;
					for (int s : slots) seen |= (s == nr);
					if (!seen) slots.push_back(nr);
				}
				for (size_t a = 0; a + 1 < slots.size(); a++)
					for (size_t b = a + 1; b < slots.size(); b++)
						if (slots[b] < slots[a]) std::swap(slots[a], slots[b]);

				for (int slot : slots) {
					uint64_t matPtr = 0;
					if (matb && slot >= 0 && slot < totcol && file.payloadAvailable(*matb) >= 8 * size_t(slot + 1))
						std::memcpy(&matPtr, file.payload(*matb) + 8 * size_t(slot), 8);
					acr_import_mesh m{};
					m.vertex_begin = uint32_t(out.positions.size() / 3);
					m.face_begin = uint32_t(out.indices.size() / 3);
					m.material_index = materialFor(matPtr);

					struct Key { uint32_t b[6]; bool operator<(const Key &k) const { return std::memcmp(b, k.b, sizeof(b)) < 0; } };
					std::map<Key, uint32_t> seen;
					auto corner = [&](uint32_t vi) -> uint32_t {
						const uint8_t *v = vp + vs->size * size_t(vi);
						float co[3], no[3];
						for (uint32_t k = 0; k < 3; k++) {
							co[k] = file.read<float>(v, *vs, "co", k);
							no[k] = float(file.read<int16_t>(v, *vs, "no", k)) / 32767.0f;
						}
						Key key;
						std::memcpy(key.b, co, 12);
						std::memcpy(key.b + 3, no, 12);
						auto it = seen.find(key);
						if (it != seen.end()) return it->second;
						const uint32_t idx = uint32_t(out.positions.size() / 3) - m.vertex_begin;
						for (int k = 0; k < 3; k++) { out.positions.push_back(co[k]); out.normals.push_back(no[k]); }
						seen.emplace(key, idx);
						return idx;
					};
					for (int p = 0; p < totpoly; p++) {
						const uint8_t *poly = pp + ps->size * size_t(p);
						if (file.read<int16_t>(poly, *ps, "mat_nr") != slot) continue;
						const int ls0 = file.read<int32_t>(poly, *ps, "loopstart"), n = file.read<int32_t>(poly, *ps, "totloop");
						if (n < 3 || ls0 < 0 || ls0 + n > totloop) continue;
						std::vector<uint32_t> c;
						for (int k = 0; k < n; k++) {
							const uint32_t vi = file.read<uint32_t>(lp + ls->size * size_t(ls0 + k), *ls, "v");
							if (vi >= uint32_t(totvert)) { error = "mesh '" + o.name + "': loop vertex out of range"; return false; }
							c.push_back(corner(vi));
						}
						for (int k = 1; k + 1 < n; k++) {
							out.indices.push_back(c[0]);
							out.indices.push_back(c[size_t(k)]);
							out.indices.push_back(c[size_t(k) + 1]);
						}
					}
					m.vertex_count = uint32_t(out.positions.size() / 3) - m.vertex_begin;
					m.face_count = uint32_t(out.indices.size() / 3) - m.face_begin;
					if (m.face_count == 0) continue;
					meshIds.push_back(uint32_t(out.meshes.size()));
					out.meshes.push_back(m);
				}
				return true;
			}

			void convertLamp(const Obj &o)
			{
				const blend::Block *lb = file.blockAt(o.data);
				const blend::Struct *ls = lb ? file.structOf(*lb) : nullptr;
				if (!lb || !ls || ls->name != "Lamp") return;
				const uint8_t *p = file.payload(*lb);
				acr_import_light l{};
				copyName(l.name, o.name);
				const int type = file.read<int16_t>(p, *ls, "type");
				const float energy = file.read<float>(p, *ls, "energy");
				const float rgb[3] = {file.read<float>(p, *ls, "r"), file.read<float>(p, *ls, "g"), file.read<float>(p, *ls, "b")};
				for (int k = 0; k < 3; k++) l.ambient[k] = l.diffuse[k] = l.specular[k] = rgb[k] * energy;
				l.att_constant = 0.0f; l.att_linear = 1.0f; l.att_quadratic = 0.0f;
				l.inner_cone = l.outer_cone = 6.28318530717958647692f;
				l.type = ACR_LIGHT_POINT;
				if (type == 1) { // SUN
					l.type = ACR_LIGHT_DIRECTIONAL;
					l.direction[2] = -1.0f;
				} else if (type == 2) { // SPOT
					l.type = ACR_LIGHT_SPOT;
					l.direction[2] = -1.0f;
					l.inner_cone = l.outer_cone = file.read<float>(p, *ls, "spotsize");
				}
				out.lights.push_back(l);
			}

			void convertCamera(const Obj &o)
			{
				if (haveCamera) return;
				haveCamera = true;
				acr_import_camera c{};
				copyName(c.name, o.name);
				c.aspect = 0.0f;
				c.hfov = 0.25f * 3.1415926535897932384626433832795f;
				c.up[1] = 1.0f;
				c.look_at[2] = -1.0f;
				out.camera = c;
			}

			bool emitNode(size_t oi, int parentNode, const math::mat4 *parentObmat)
			{
				const Obj &o = objs[oi];
				acr_import_node n{};
				n.parent = parentNode;
				copyName(n.name, o.name);
				const math::mat4 local = parentObmat ? math::inverse(*parentObmat) * o.obmat : o.obmat;
				for (int c = 0; c < 4; c++) {
					n.transform[4 * c] = local.c[c].x; n.transform[4 * c + 1] = local.c[c].y;
					n.transform[4 * c + 2] = local.c[c].z; n.transform[4 * c + 3] = local.c[c].w;
				}
				std::vector<uint32_t> meshIds;
				if (o.type == 1 && !convertMesh(o, meshIds)) return false;
				if (o.type == 10) convertLamp(o);
				if (o.type == 11) convertCamera(o);
				n.mesh_begin = uint32_t(out.nodeMeshIds.size());
				n.mesh_count = uint32_t(meshIds.size());
				for (uint32_t id : meshIds) out.nodeMeshIds.push_back(id);
				const int self = int(out.nodes.size());
				out.nodes.push_back(n);
				for (size_t c : o.children)
					if (!emitNode(c, self, &o.obmat)) return false;
				return true;
			}
		};
	} // namespace

	bool loadBlend(const char *path, ImportScene &out, std::string &error)
	{
		blend::File file;
		if (!file.open(path, error)) return false;
		out = ImportScene();

		const blend::Block *sc = nullptr;
		for (const blend::Block &b : file.blocks())
			if (std::strcmp(b.code, "SC") == 0) { sc = &b; break; }
		const blend::Struct *ss = sc ? file.structOf(*sc) : nullptr;
		const blend::Struct *bs = file.structByName("Base");
		const blend::Struct *os = file.structByName("Object");
		if (!sc || !ss || !bs || !os) {
			error = "no scene in file";
			return false;
		}

		Converter cv(file, out);
		std::map<uint64_t, size_t> objIndex;
		uint64_t basePtr = file.readPointer(file.payload(*sc), *ss, "base.first");
		for (int guard = 0; basePtr && guard < 1000000; guard++) {
			const blend::Block *bb = file.blockAt(basePtr);
			if (!bb) break;
			const uint8_t *bp = file.payload(*bb);
			const uint64_t objPtr = file.readPointer(bp, *bs, "object");
			const blend::Block *ob = file.blockAt(objPtr);
			if (ob && file.structOf(*ob) && file.structOf(*ob)->name == "Object" && !objIndex.count(objPtr)) {
				const uint8_t *op = file.payload(*ob);
				Obj o;
				o.ptr = objPtr;
				o.parent = file.readPointer(op, *os, "parent");
				o.data = file.readPointer(op, *os, "data");
				o.type = file.read<int16_t>(op, *os, "type");
				const std::string idname = file.readString(op, *os, "id.name");
				o.name = idname.size() > 2 ? idname.substr(2) : idname;
				for (uint32_t c = 0; c < 4; c++)
					o.obmat.c[c] = math::vec4(file.read<float>(op, *os, "obmat", 4 * c), file.read<float>(op, *os, "obmat", 4 * c + 1),
					                          file.read<float>(op, *os, "obmat", 4 * c + 2), file.read<float>(op, *os, "obmat", 4 * c + 3));
				objIndex.emplace(objPtr, cv.objs.size());
				cv.objs.push_back(o);
			}
			basePtr = file.readPointer(bp, *bs, "next");
		}
		std::vector<size_t> roots;
		for (size_t i = 0; i < cv.objs.size(); i++) {
			auto it = cv.objs[i].parent ? objIndex.find(cv.objs[i].parent) : objIndex.end();
			if (it == objIndex.end()) roots.push_back(i);
			else cv.objs[it->second].children.push_back(i);
		}

		acr_import_node root{};
		root.parent = -1;
		copyName(root.name, "<BlenderRoot>");
		const math::mat4 I;
		for (int c = 0; c < 4; c++) {
			root.transform[4 * c] = I.c[c].x; root.transform[4 * c + 1] = I.c[c].y; root.transform[4 * c + 2] = I.c[c].z; root.transform[4 * c + 3] = I.c[c].w;
		}
		out.nodes.push_back(root);
		for (size_t r : roots)
			if (!cv.emitNode(r, 0, nullptr)) {
				error = cv.error;
				return false;
			}
		if (!cv.haveCamera) {
			error = "scene has no camera"; // the reference dereferences mCameras[0] unconditionally (scene.cu:104)
			return false;
		}
		return true;
	}
} // namespace acr
