// Scene::flatten -- host scene -> structure-of-arrays image for
// acr_hip_upload_scene.  Pure copying: nothing here computes a float, except
// that vertex data is gathered through Face::indices so the kernels read one
// contiguous record per triangle (the reference chases vertices[indices[k]]
// on the device, geometry.h:126-133).
#include "scene.h"

namespace acr
{
	namespace
	{
		void pushMat4(std::vector<float> &v, const math::mat4 &m)
		{
			for (int c = 0; c < 4; c++) {
				v.push_back(m.c[c].x); v.push_back(m.c[c].y); v.push_back(m.c[c].z); v.push_back(m.c[c].w);
			}
		}
		void pushMat3(std::vector<float> &v, const math::mat3 &m)
		{
			for (int c = 0; c < 3; c++) {
				v.push_back(m.c[c].x); v.push_back(m.c[c].y); v.push_back(m.c[c].z);
			}
		}
		void pushVec3(std::vector<float> &v, const math::vec3 &a)
		{
			v.push_back(a.x); v.push_back(a.y); v.push_back(a.z);
		}
		template <class NodeVec>
		void pushNodes(const NodeVec &nodes, std::vector<uint32_t> &start, std::vector<uint32_t> &end, std::vector<float> &left,
		               std::vector<float> &right, std::vector<uint8_t> &flags)
		{
			for (const auto &n : nodes) {
				start.push_back(n.start);
				end.push_back(n.end);
				left.push_back(n.left);
				right.push_back(n.right);
				flags.push_back(uint8_t((n.axis & ACR_NODE_AXIS_MASK) | (n.isLeaf ? ACR_NODE_LEAF : 0)));
			}
		}
	} // namespace

	const acr_scene_soa &Scene::flatten() const
	{
		if (arrays) return arrays->view;
		arrays = new SceneArrays();
		SceneArrays &a = *arrays;

		pushNodes(objects.nodes(), a.snodeStart, a.snodeEnd, a.snodeLeft, a.snodeRight, a.snodeFlags);
		const vector<Object> &objs = objects.items();
		for (size_t i = 0; i < objs.size(); i++) {
			const Object &o = objs[i];
			pushMat4(a.objTransform, o.globalTransform);
			pushMat4(a.objInvTransform, o.globalInverseTransform);
			pushMat3(a.objNormal, o.globalNormalTransform);
			pushMat3(a.objInvNormal, o.globalInverseNormalTransform);
			pushVec3(a.objWorldMin, o.boundingBox.min);
			pushVec3(a.objWorldMax, o.boundingBox.max);
			a.objMeshBegin.push_back(uint32_t(a.objMeshIds.size()));
			a.objMeshCount.push_back(uint32_t(o.meshes.size()));
			for (size_t k = 0; k < o.meshes.size(); k++)
				a.objMeshIds.push_back(uint32_t(o.meshes[k]));
		}

		for (size_t m = 0; m < meshes.size(); m++) {
			const Mesh &mesh = meshes[m];
			const BIH<Face> &bih = mesh.getFaces();
			pushVec3(a.meshMin, mesh.boundingBox.min);
			pushVec3(a.meshMax, mesh.boundingBox.max);
			a.meshMaterial.push_back(mesh.getMaterialIndex());
			a.meshDepth.push_back(uint32_t(bih.maxDepth()));
			a.meshNodeBegin.push_back(uint32_t(a.mnodeStart.size()));
			a.meshTriBegin.push_back(uint32_t(a.triPrimId.size()));
			pushNodes(bih.nodes(), a.mnodeStart, a.mnodeEnd, a.mnodeLeft, a.mnodeRight, a.mnodeFlags);
			const vector<Face> &faces = bih.items();
			const vector<Vertex> &verts = mesh.getVertices();
			const std::vector<uint32_t> &order = bih.itemOrder();
			for (size_t f = 0; f < faces.size(); f++) {
				const Vertex &v0 = verts[faces[f].indices[0]];
				const Vertex &v1 = verts[faces[f].indices[1]];
				const Vertex &v2 = verts[faces[f].indices[2]];
				pushVec3(a.triP0, v0.position); pushVec3(a.triP1, v1.position); pushVec3(a.triP2, v2.position);
				pushVec3(a.triN0, v0.normal); pushVec3(a.triN1, v1.normal); pushVec3(a.triN2, v2.normal);
				a.triPrimId.push_back(mesh.getFirstPrimitive() + order[f]);
			}
		}

		for (size_t i = 0; i < materials.size(); i++) {
			pushVec3(a.matAmbient, materials[i].ambient);
			pushVec3(a.matDiffuse, materials[i].diffuse);
			pushVec3(a.matSpecular, materials[i].specular);
			a.matRefractiveIndex.push_back(materials[i].refractiveIndex);
		}
		for (size_t i = 0; i < lights.size(); i++) {
			const Light &l = lights[i];
			a.lightType.push_back(l.type == Light::POINT ? ACR_LIGHT_POINT : (l.type == Light::DIRECTIONAL ? ACR_LIGHT_DIRECTIONAL : ACR_LIGHT_SPOT));
			pushVec3(a.lightDiffuse, l.diffuse);
			pushVec3(a.lightPosition, l.position);
			pushVec3(a.lightDirection, l.direction);
			a.lightAttenuation.push_back(l.attConstant); a.lightAttenuation.push_back(l.attLinear); a.lightAttenuation.push_back(l.attQuadratic);
			a.lightCone.push_back(l.innerConeAngle); a.lightCone.push_back(l.outerConeAngle);
		}

		acr_scene_soa &v = a.view;
		v.scene_depth = uint32_t(objects.maxDepth());
		const float mn[3] = {boundingBox.min.x, boundingBox.min.y, boundingBox.min.z};
		const float mx[3] = {boundingBox.max.x, boundingBox.max.y, boundingBox.max.z};
		for (int k = 0; k < 3; k++) { v.scene_min[k] = mn[k]; v.scene_max[k] = mx[k]; }
		v.snode_start = a.snodeStart.data(); v.snode_end = a.snodeEnd.data();
		v.snode_left = a.snodeLeft.data(); v.snode_right = a.snodeRight.data(); v.snode_flags = a.snodeFlags.data();
		v.num_objects = uint32_t(objs.size());
		v.obj_transform = a.objTransform.data(); v.obj_inv_transform = a.objInvTransform.data();
		v.obj_normal_transform = a.objNormal.data(); v.obj_inv_normal_transform = a.objInvNormal.data();
		v.obj_world_min = a.objWorldMin.data(); v.obj_world_max = a.objWorldMax.data();
		v.obj_mesh_begin = a.objMeshBegin.data(); v.obj_mesh_count = a.objMeshCount.data();
		v.num_obj_mesh_ids = uint32_t(a.objMeshIds.size()); v.obj_mesh_ids = a.objMeshIds.data();
		v.num_meshes = uint32_t(meshes.size());
		v.mesh_min = a.meshMin.data(); v.mesh_max = a.meshMax.data();
		v.mesh_material = a.meshMaterial.data(); v.mesh_depth = a.meshDepth.data();
		v.mesh_node_begin = a.meshNodeBegin.data(); v.mesh_tri_begin = a.meshTriBegin.data();
		v.num_mesh_nodes = uint32_t(a.mnodeStart.size());
		v.mnode_start = a.mnodeStart.data(); v.mnode_end = a.mnodeEnd.data();
		v.mnode_left = a.mnodeLeft.data(); v.mnode_right = a.mnodeRight.data(); v.mnode_flags = a.mnodeFlags.data();
		v.num_triangles = uint32_t(a.triPrimId.size());
		v.tri_p0 = a.triP0.data(); v.tri_p1 = a.triP1.data(); v.tri_p2 = a.triP2.data();
		v.tri_n0 = a.triN0.data(); v.tri_n1 = a.triN1.data(); v.tri_n2 = a.triN2.data();
		v.tri_prim_id = a.triPrimId.data();
		v.num_materials = uint32_t(materials.size());
		v.mat_ambient = a.matAmbient.data(); v.mat_diffuse = a.matDiffuse.data(); v.mat_specular = a.matSpecular.data();
		v.mat_refractive_index = a.matRefractiveIndex.data();
		v.num_lights = uint32_t(lights.size());
		v.light_type = a.lightType.data(); v.light_diffuse = a.lightDiffuse.data(); v.light_position = a.lightPosition.data();
		v.light_direction = a.lightDirection.data(); v.light_attenuation = a.lightAttenuation.data(); v.light_cone = a.lightCone.data();
		return v;
	}
} // namespace acr
