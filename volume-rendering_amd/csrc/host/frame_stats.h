// frame_stats.h — frame-time statistics of the headless driver: one accumulator per (benchmark configuration, renderer),
// reported the way the reference's profiler reports (VolumeRendering/Profiler.cpp:80-114): averages and maxima only once a
// cell holds 8 samples, "N/A" otherwise.  The timed region is Renderer::render_volume() as in the reference
// (VolR.cpp:109-111).  `Profiler` keeps the reference's static call names for the driver's benchmark loop.
#pragma once

#include <chrono>
#include <stdio.h>
#include <vector>

namespace volr {

class FrameStats {
	public:
		struct Cell {
			unsigned samples = 0;
			float worst_ms = 0.0f;
			double total_ms = 0.0;
			bool reportable() const { return samples >= 8; }               // MIN_SAMPLE_STAT
			double mean_ms() const { return samples ? total_ms / samples : 0.0; }
		};
		FrameStats(int configs, int renderers) : renderers_(renderers), cells_((size_t) configs * renderers) {}
		void clear(int config);
		void add(int config, int renderer, float ms);
		const Cell &cell(int config, int renderer) const { return cells_[(size_t) config * renderers_ + renderer]; }
		int renderers() const { return renderers_; }
		int configs() const { return (int) (cells_.size() / renderers_); }
		// one table line each, in the reference's column format ("%9s" label, "%8..." cells, comma separated)
		void print_counts(FILE *out, int config) const;
		void print_means(FILE *out, int config) const;
		void print_worst(FILE *out, int config) const;
	private:
		int renderers_;
		std::vector<Cell> cells_;
};

constexpr int PROFILER_RENDERERS = 2;     // this project ships two back-ends: HIP nearest, HIP trilinear

class Profiler {
	public:
		static void init();
		static void reset_config(int config);
		static void start(int renderer);
		static float stop();                  // wall clock around a synchronous render_volume(); -1 without a start()
		static void print_samples(FILE *out, int config) { table().print_counts(out, config); }
		static void print_avg(FILE *out, int config) { table().print_means(out, config); }
		static void print_max(FILE *out, int config) { table().print_worst(out, config); }
		static float time_ms;
	private:
		static FrameStats &table();
};

}  // namespace volr
