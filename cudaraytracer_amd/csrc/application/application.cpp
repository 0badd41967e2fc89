#include "application.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <stdexcept>

#include "../loader/import_scene.h"

namespace acr
{
	namespace
	{
		const float movSpeed = 0.5f;           // application.cpp:11
		const float rotSpeed = 1.0f / 1000.0f; // application.cpp:45

		Scene makeScene(const Application::Args &a)
		{
			if (a.soupTriangles) {
				ImportScene imp;
				makeSoup(a.soupTriangles, 1, 0.1f, 10.0f, imp);
				const acr_import_scene view = imp.view();
				return Scene(view, a.scene.sceneBihDepth, a.scene.meshBihDepth);
			}
			if (!a.scene.filePath) throw std::runtime_error("no scene file given");
			return Scene(a.scene);
		}
	} // namespace

	Application::Application(const Args &a) : renderer(a.renderer), scene(makeScene(a)), args(a)
	{
		renderer.setMaxBounces(a.maxBounces);
		renderer.setFixedSeed(a.seed);
	}

	void Application::key(unsigned char k)
	{
		switch (k) {
		case 'w': renderer.moveCamera(math::vec2(0, 1) * movSpeed, math::vec2(0, 0)); break;
		case 'a': renderer.moveCamera(math::vec2(-1, 0) * movSpeed, math::vec2(0, 0)); break;
		case 's': renderer.moveCamera(math::vec2(0, -1) * movSpeed, math::vec2(0, 0)); break;
		case 'd': renderer.moveCamera(math::vec2(1, 0) * movSpeed, math::vec2(0, 0)); break;
		default: break;
		}
	}

	void Application::drag(float dx, float dy)
	{
		math::vec2 delta = math::vec2(dx, dy) * rotSpeed;
		delta.x *= -1;
		renderer.moveCamera(math::vec2(0, 0), delta);
	}

	void Application::start()
	{
		srand(unsigned(time(nullptr))); // application.cpp:76
		std::cout << "Starting application..." << std::endl;
		renderer.loadScene(scene);
		const std::string &ev = args.events;
		for (size_t i = 0; i < ev.size(); i++) {
			const char c = ev[i];
			if (c == '.') {
				renderer.render();
				frames++;
			} else if (c == 'r') {
				float dx = 0, dy = 0;
				int used = 0;
				if (std::sscanf(ev.c_str() + i + 1, "%f,%f;%n", &dx, &dy, &used) != 2 || used == 0)
					throw std::runtime_error("bad drag event at position " + std::to_string(i) + " (want r<dx>,<dy>;)");
				drag(dx, dy);
				i += size_t(used);
			} else {
				key(static_cast<unsigned char>(c));
			}
		}
		if (!args.out.empty()) renderer.save(args.out);
	}
} // namespace acr

static void usage(const char *prog)
{
	std::fprintf(stderr,
	             "usage: %s <scene.blend> | --soup N   [--size WxH] [--events SCRIPT] [--frames N] [--out image.ppm|.pfm]\n"
	             "          [--bounces N] [--seed S] [--mesh-depth D] [--device G | --devices G0,G1,...]\n"
	             "  --devices: every frame across these GPUs (bands of the frame per GPU, the frame assembled on G0); a GPU may be named twice\n"
	             "  SCRIPT: w a s d = key presses, r<dx>,<dy>; = mouse drag, . = render one frame;  --frames N appends N dots\n",
	             prog);
}

// application.cpp:87-106: 800x600 by default, scene file from the command line
int main(int argc, char **argv)
{
	acr::Application::Args args;
	args.renderer.title = "HipRenderer";
	args.renderer.pos.x = 0;
	args.renderer.pos.y = 0;
	args.renderer.dim.x = 800;
	args.renderer.dim.y = 600;
	args.renderer.dim.z = 1;
	args.frameRate = 60;
	args.scene.filePath = nullptr;
	args.events.clear();
	unsigned frames = 0;
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto next = [&]() -> const char * {
			if (i + 1 >= argc) { usage(argv[0]); std::exit(2); }
			return argv[++i];
		};
		if (a == "--size") {
			unsigned w = 0, h = 0;
			if (std::sscanf(next(), "%ux%u", &w, &h) != 2 || !w || !h) { usage(argv[0]); return 2; }
			args.renderer.dim.x = w;
			args.renderer.dim.y = h;
		} else if (a == "--events") args.events += next();
		else if (a == "--frames") frames = unsigned(std::atoi(next()));
		else if (a == "--out") args.out = next();
		else if (a == "--bounces") args.maxBounces = unsigned(std::atoi(next()));
		else if (a == "--seed") args.seed = std::atoll(next());
		else if (a == "--soup") args.soupTriangles = unsigned(std::atoi(next()));
		else if (a == "--mesh-depth") args.scene.meshBihDepth = size_t(std::atoi(next()));
		else if (a == "--device") args.renderer.device = std::atoi(next());
		else if (a == "--devices") {
			args.renderer.devices.clear();
			for (const char *q = next(); *q;) {
				char *end = nullptr;
				const long g = std::strtol(q, &end, 10);
				if (end == q) { usage(argv[0]); return 2; }
				args.renderer.devices.push_back(int(g));
				q = *end == ',' ? end + 1 : end;
			}
		}
		else if (a.size() && a[0] != '-') args.scene.filePath = argv[i];
		else { usage(argv[0]); return 2; }
	}
	args.events.append(frames, '.');
	if (args.events.empty()) args.events = ".";
	if (!args.scene.filePath && !args.soupTriangles) { usage(argv[0]); return 2; }
	try {
		acr::Application app(args);
		app.start();
		std::cout << app.framesRendered() << " frame(s) rendered" << (args.out.empty() ? "" : ", last one written to " + args.out) << std::endl;
	} catch (const std::exception &e) {
		std::cerr << "error: " << e.what() << std::endl; // the reference exits with EXIT_FAILURE from inside its loaders
		return EXIT_FAILURE;
	}
	return 0;
}
