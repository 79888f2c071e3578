// Profiler.h — host mirror of the reference's per-(configuration, renderer) timing statistics
// (VolumeRendering/Profiler.h:14-38, Profiler.cpp): sum / max / sample count, averages printed only from 8 samples on.
// The timed region is Renderer::render_volume() as in the reference (VolR.cpp:109-111): clear + kernel (+ copy-out).
#pragma once

#include <stdio.h>

namespace volr {

constexpr int MAX_CONFIG_COUNT = 100;
constexpr int MIN_SAMPLE_STAT = 8;
constexpr int PROFILER_RENDERERS = 2;     // this project ships two back-ends: HIP nearest, HIP trilinear

struct Stat {
	unsigned int samples;
	float time_max;
	double time_sum;
};

class Profiler {
	public:
		static void init();
		static void reset_config(int config);
		static void start(int renderer);
		static float stop();                 // wall clock around a synchronous render_volume(); -1 without a start()
		static void record(float ms);        // alternative: feed a device-side time (vr_hip_timing) for the started renderer
		static void print_samples(FILE *out, int config);
		static void print_avg(FILE *out, int config);
		static void print_max(FILE *out, int config);
		static const Stat &stat(int config, int renderer) { return statistics[config][renderer]; }
		static float time_ms;
	private:
		static Stat statistics[MAX_CONFIG_COUNT][PROFILER_RENDERERS];
		static int current_config, current_renderer;
};

}  // namespace volr
