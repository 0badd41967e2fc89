// acr::Application (reference src/application/application.h:18-45), headless.
//
// Same shape -- Args{renderer, frameRate, scene}, start() -- and the same reactions to input
// (application.cpp:11-58: w/a/s/d move the camera by movSpeed, a mouse drag turns it by rotSpeed per pixel),
// but where the reference hands control to glutMainLoop() and reads events from the window, this one reads them
// from a script and writes the last frame to a file:
//
//     w a s d          one key press each (application.cpp:16-27)
//     r<dx>,<dy>;      a mouse drag by (dx, dy) pixels with the button down (application.cpp:47-58)
//     .                render one frame (the reference renders one per display callback)
//     anything else    ignored (white space separates)
#ifndef ACR_APPLICATION_APPLICATION_H
#define ACR_APPLICATION_APPLICATION_H

#include <string>

#include "../math/math.h"
#include "../renderer/renderer.h"
#include "../scene/scene.h"

namespace acr
{
	class Application
	{
	public:
		struct Args
		{
			Renderer::Args renderer;
			uint8_t frameRate = 60; // kept for the shape of the reference's Args; a script has no clock
			Scene::Args scene;      // filePath may be null when `soupTriangles` is set
			uint32_t soupTriangles = 0; // > 0: a procedural triangle soup instead of a file (seed 1, edges 0.1, extent 10)
			std::string events = "."; // the script
			std::string out;          // .ppm / .pfm written after the last event; empty = none
			uint32_t maxBounces = 5;  // MAX_BOUNCES (renderer.cu:10)
			long long seed = -1;      // >= 0: frame k is seeded seed + k instead of rand() (renderer.cu:421)
		};

		explicit Application(const Args &args);

		void start(); // application.cpp:74-82 with the script in place of glutMainLoop()

		Renderer renderer;
		Scene scene;

		uint32_t framesRendered() const { return frames; }

	private:
		void key(unsigned char k);        // keyboardCB, application.cpp:12-33
		void drag(float dx, float dy);    // mouseMoveCB with the button down, application.cpp:47-58

		Args args;
		uint32_t frames = 0;
	};
} // namespace acr

#endif
