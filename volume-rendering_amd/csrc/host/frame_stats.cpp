// frame_stats.cpp — see frame_stats.h.
#include "frame_stats.h"

namespace volr {

void FrameStats::clear(int config) {
	for (int r = 0; r < renderers_; r++)
		cells_[(size_t) config * renderers_ + r] = Cell();
}

void FrameStats::add(int config, int renderer, float ms) {
	Cell &c = cells_[(size_t) config * renderers_ + renderer];
	c.samples++;
	c.total_ms += ms;
	if (ms > c.worst_ms) c.worst_ms = ms;
}

void FrameStats::print_counts(FILE *out, int config) const {
	fprintf(out, "%9s", "Samples,");
	for (int r = 0; r < renderers_; r++)
		fprintf(out, "%8u%s", cell(config, r).samples, r + 1 < renderers_ ? "," : "\n");
}

void FrameStats::print_means(FILE *out, int config) const {
	fprintf(out, "%9s", "Avg(ms),");
	for (int r = 0; r < renderers_; r++) {
		const Cell &c = cell(config, r);
		if (c.reportable()) fprintf(out, "%8.2f", c.mean_ms());
		else fprintf(out, "%8s", "N/A");
		fputs(r + 1 < renderers_ ? "," : "\n", out);
	}
}

void FrameStats::print_worst(FILE *out, int config) const {
	fprintf(out, "%9s", "Max(ms),");
	for (int r = 0; r < renderers_; r++) {
		const Cell &c = cell(config, r);
		if (c.reportable()) fprintf(out, "%8.2f", c.worst_ms);
		else fprintf(out, "%8s", "N/A");
		fputs(r + 1 < renderers_ ? "," : "\n", out);
	}
}

// ---- the reference's static call names --------------------------------------------------------------------------------

namespace {
int active_config = 0, active_renderer = -1;
std::chrono::steady_clock::time_point started;
}

float Profiler::time_ms = -1;

FrameStats &Profiler::table() {
	static FrameStats stats(100, PROFILER_RENDERERS);                   // MAX_CONFIG_COUNT
	return stats;
}

void Profiler::init() {
	for (int c = 0; c < table().configs(); c++) table().clear(c);
	active_config = 0;
	active_renderer = -1;
	time_ms = -1;
}

void Profiler::reset_config(int config) {
	if (config < 0) config = 0;
	if (config >= table().configs()) config = table().configs() - 1;
	table().clear(config);
	active_config = config;
	active_renderer = -1;
	time_ms = -1;
}

void Profiler::start(int renderer) {
	active_renderer = renderer;
	started = std::chrono::steady_clock::now();
}

float Profiler::stop() {
	if (active_renderer < 0 || active_renderer >= table().renderers())
		return -1;
	time_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - started).count();
	table().add(active_config, active_renderer, time_ms);
	active_renderer = -1;
	return time_ms;
}

}  // namespace volr
