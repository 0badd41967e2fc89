// acr::Camera (reference src/scene/camera.h:7-18, scene.cu:294-301)
#ifndef ACR_SCENE_CAMERA_H
#define ACR_SCENE_CAMERA_H

#include "../../../include/acr_host.h"
#include "../math/math.h"

namespace acr
{
	class Camera
	{
	public:
		Camera() : aspectRatio(0), horizontalFOV(0) {}
		explicit Camera(const acr_import_camera &cam)
			: aspectRatio(cam.aspect), horizontalFOV(cam.hfov), position(cam.position[0], cam.position[1], cam.position[2]),
			  forward(cam.look_at[0], cam.look_at[1], cam.look_at[2]), up(cam.up[0], cam.up[1], cam.up[2])
		{
		}
		float aspectRatio;
		float horizontalFOV;
		math::vec3 position, forward, up;
	};
} // namespace acr

#endif
