// Profiler.cpp — see Profiler.h (behaviour of VolumeRendering/Profiler.cpp; output format of its print_* methods).
#include "Profiler.h"

#include <chrono>

namespace volr {

Stat Profiler::statistics[MAX_CONFIG_COUNT][PROFILER_RENDERERS];
int Profiler::current_config = 0, Profiler::current_renderer = -1;
float Profiler::time_ms = -1;

static std::chrono::steady_clock::time_point t_start;

void Profiler::init() {
	for (int c = 0; c < MAX_CONFIG_COUNT; c++)
		reset_config(c);
	current_config = 0;
}

void Profiler::reset_config(int config) {
	if (config < 0) config = 0;
	if (config >= MAX_CONFIG_COUNT) config = MAX_CONFIG_COUNT - 1;
	for (int r = 0; r < PROFILER_RENDERERS; r++)
		statistics[config][r] = Stat{ 0, 0.0f, 0.0 };
	current_renderer = -1;
	time_ms = -1;
	current_config = config;
}

void Profiler::start(int renderer) {
	current_renderer = renderer;
	t_start = std::chrono::steady_clock::now();
}

void Profiler::record(float ms) {
	if (current_renderer < 0 || current_renderer >= PROFILER_RENDERERS)
		return;
	Stat &s = statistics[current_config][current_renderer];
	s.samples++;
	s.time_sum += ms;
	if (ms > s.time_max) s.time_max = ms;
	time_ms = ms;
	current_renderer = -1;
}

float Profiler::stop() {
	if (current_renderer < 0)
		return -1;
	const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
	record(ms);
	return ms;
}

void Profiler::print_samples(FILE *out, int config) {
	fprintf(out, "%9s", "Samples,");
	for (int r = 0; r < PROFILER_RENDERERS; r++)
		fprintf(out, "%8u%s", statistics[config][r].samples, r != PROFILER_RENDERERS - 1 ? "," : "");
	fprintf(out, "\n");
}

void Profiler::print_avg(FILE *out, int config) {
	fprintf(out, "%9s", "Avg(ms),");
	for (int r = 0; r < PROFILER_RENDERERS; r++) {
		const Stat &s = statistics[config][r];
		if (s.samples >= MIN_SAMPLE_STAT) fprintf(out, "%8.2f", s.time_sum / s.samples);
		else fprintf(out, "%8s", "N/A");
		if (r != PROFILER_RENDERERS - 1) fprintf(out, ",");
	}
	fprintf(out, "\n");
}

void Profiler::print_max(FILE *out, int config) {
	fprintf(out, "%9s", "Max(ms),");
	for (int r = 0; r < PROFILER_RENDERERS; r++) {
		const Stat &s = statistics[config][r];
		if (s.samples >= MIN_SAMPLE_STAT) fprintf(out, "%8.2f", s.time_max);
		else fprintf(out, "%8s", "N/A");
		if (r != PROFILER_RENDERERS - 1) fprintf(out, ",");
	}
	fprintf(out, "\n");
}

}  // namespace volr
