// volr_bench — headless command-line driver over the host mirror: the reference's start-up sequence
// (VolR.cpp:352-442) without GLUT/GLUI, its benchmark loop and configuration matrix (VolR.cpp:225-321), its profiler
// summary table (VolR.cpp:200-223) and, for eyeballing, a PPM writer for single frames.  There is no interactive mode.
//
//   volr_bench [-h] [-f <file.pvm|.raw>] [-raw <w> <h> <d> [<bytes>]] [-synthetic <n>] [-dir <datasets>] [-r <id>]
//              [-s <width> <height>] [-d <device>] [-devices <a,b,..>] [-b|-bg] [-pose <ax> <ay> <az> <dist>] [-persp] [-o <frame.ppm>]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "ModelBase.h"
#include "frame_stats.h"
#include "RaycasterBase.h"
#include "camera.h"

using namespace volr;

namespace {

const float MAX_BENCH_SAMPLE = 7500;         // VolR.cpp:26: a renderer is dropped from a configuration after a 7.5 s frame

int config = 0, device = 0, renderer_id = 1;
std::vector<std::string> row_names = { "Interactive" };     // profiler row labels: row 0 is the interactive / single-frame row
std::vector<int> device_list;       // -devices a,b,c: one frame split over several GPUs (vr_hip_multi_*)
std::string dataset_dir = ".";
unsigned synthetic_n = 0;
std::vector<unsigned char> synthetic_voxels;
std::vector<uchar4> frame;
HipRenderer *renderers[PROFILER_RENDERERS] = { nullptr, nullptr };

void print_usage() {
	printf("volr_bench - headless MI355X volume raycaster driver (benchmark matrix of VolR -b)\n\n"
	       "  -h : this help\n  -f <file> : volume data, .pvm or .raw (with -raw <w> <h> <d> [<bytes per voxel>])\n"
	       "  -synthetic <n> : n^3 synthetic shell volume (stands in for missing datasets in benchmark mode)\n"
	       "  -dir <path> : directory searched for <Name>.pvm in benchmark mode (default .)\n"
	       "  -r <id> : renderer, 0 = HIP nearest (CPURenderer semantics), 1 = HIP trilinear (GPURenderer4 semantics)\n"
	       "  -s <width> <height> : viewport, 128..2048 like the reference\n  -d <device> : GPU index\n"
	       "  -devices <a,b,...> : split every frame over these GPUs (interleaved bands gathered on the first one over xGMI)\n"
	       "  -b | -bg : benchmark mode\n  -pose <ax> <ay> <az> <dist> [-persp] -o <frame.ppm> : render one frame to a PPM file\n");
}

// SURVEY §8(d) "shell": deterministic integer-only test volume (same generator as the GPU kernel and the oracle)
void make_shell(unsigned n, std::vector<unsigned char> *out) {
	out->resize((size_t) n * n * n);
	const long long N = n;
	for (long long z = 0; z < N; z++)
		for (long long y = 0; y < N; y++)
			for (long long x = 0; x < N; x++) {
				const unsigned long long idx = ((unsigned long long) z * N + y) * N + x;
				unsigned h = (unsigned) (idx ^ (idx >> 32)) + 1u * 0x9E3779B9u;
				h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
				const long long ax = 2 * x + 1 - N, ay = 2 * y + 1 - N, az = 2 * z + 1 - N;
				long long t = 1000 * (ax * ax + ay * ay + az * az) / (N * N) - 360;
				if (t < 0) t = -t;
				long long shell = 255 - t * 255 / 240;
				if (shell < 0) shell = 0;
				unsigned v = (unsigned) shell + (h & 15u);
				(*out)[idx] = (unsigned char) (v > 255u ? 255u : v);
			}
}

bool use_synthetic() {
	if (synthetic_n == 0) return false;
	if (synthetic_voxels.empty()) make_shell(synthetic_n, &synthetic_voxels);
	// ModelBase owns (and frees) its voxel buffer like the reference does (ModelBase.cpp:99-100): hand it a malloc'ed copy
	unsigned char *copy = (unsigned char *) malloc(synthetic_voxels.size());
	if (copy == NULL) return false;
	memcpy(copy, synthetic_voxels.data(), synthetic_voxels.size());
	if (ModelBase::volume.data != NULL) free(ModelBase::volume.data);
	ModelBase::volume.data = copy;
	ModelBase::volume.size = (unsigned) synthetic_voxels.size();
	ModelBase::volume.dims = make_ushort3((unsigned short) synthetic_n, (unsigned short) synthetic_n, (unsigned short) synthetic_n);
	return true;
}

// VolR.cpp:255-268 benchmark_load_file (+ every renderer's set_volume / set_transfer_fn)
int install_volume() {
	RaycasterBase::set_volume(ModelBase::volume);
	for (int i = 0; i < PROFILER_RENDERERS; i++) {
		if (renderers[i]->set_volume(RaycasterBase::raycaster.volume) != 0) return 1;
		renderers[i]->set_transfer_fn(RaycasterBase::raycaster);
	}
	return 0;
}

int benchmark_load_file(const char *name) {
	const std::string path = dataset_dir + "/" + name + ".pvm";
	if (ModelBase::load_model(path.c_str()) != 0) {
		printf("Error: File not found: %s\n", path.c_str());
		return 1;
	}
	return install_volume();
}

// VolR.cpp:98-113 draw_volume without the GL parts
void draw_volume() {
	if (frame.size() != (size_t) ViewBase::view.dims.x * ViewBase::view.dims.y) {
		frame.resize((size_t) ViewBase::view.dims.x * ViewBase::view.dims.y);
		for (int i = 0; i < PROFILER_RENDERERS; i++) renderers[i]->set_window_buffer(ViewBase::view);
	}
	RaycasterBase::set_view(ViewBase::view);
	Profiler::start(renderer_id);
	renderers[renderer_id]->render_volume(frame.data(), RaycasterBase::raycaster);
	Profiler::stop();
}

// VolR.cpp:225-253 benchmark_config_loop: every renderer x {orthogonal, perspective} x 4 poses at distance 2
void benchmark_config_loop() {
	static const float poses[4][3] = { { 0, 0, 0 }, { -45, -45, 0 }, { 90, 0, 0 }, { 180, 90, 0 } };
	Profiler::reset_config(config);
	for (renderer_id = 0; renderer_id < PROFILER_RENDERERS; renderer_id++) {
		ViewBase::view.perspective = false;
		bool timed_out = false;
		for (int p = 0; p < 2 && !timed_out; p++) {
			ViewBase::toggle_perspective(1);
			for (int i = 0; i < 4; i++) {
				ViewBase::set_camera_position(make_float3(poses[i][0], poses[i][1], poses[i][2]), 2);
				draw_volume();
				if (Profiler::time_ms > MAX_BENCH_SAMPLE) { timed_out = true; break; }
			}
			ViewBase::view.perspective = true;
		}
	}
	printf("%15s,", row_names[config].c_str());
	Profiler::print_avg(stdout, config);
	config++;
}

// ---- the reference's configuration matrix (VolR.cpp:270-321) as DATA: one row per configuration, one loop that runs them ----
struct BenchConfig {
	std::string name;            // row label of the profiler table (VolR.cpp:34-38)
	std::string dataset;         // "<name>.pvm" to load before the run; empty = keep the resident volume
	bool foot_study;             // part of the option / scale / ray-step studies the reference runs on "Foot"
	int esl;                     // empty-space leaping on (1) / off (0)
	float ray_threshold;         // early-ray-termination threshold
	float viewport_scale;        // 0 = the original viewport, else original * scale (ushort truncation, ViewBase.cpp:107-113)
	float ray_step_factor;       // 0 = the volume's default step, else default * factor
};

std::vector<BenchConfig> build_matrix() {
	std::vector<BenchConfig> m;
	for (const char *ds : { "Bucky", "Daisy", "VisMale", "Engine", "Foot", "Pig", "Porsche" })
		m.push_back({ ds, ds, false, 1, 0.95f, 0.0f, 0.0f });
	m.push_back({ "Foot: No optims", "", true, 0, 1.0f, 0.0f, 0.0f });
	m.push_back({ "F: ERT on", "", true, 0, 0.95f, 0.0f, 0.0f });
	m.push_back({ "F: ERT+ESL on", "", true, 1, 0.95f, 0.0f, 0.0f });
	// The reference steps both studies by repeated fp32 addition of 0.1 (0.9, 0.79999995, ... 0.29999992; 1.1 ... 1.7000002);
	// the accumulated values decide the truncated viewport sizes (2048 * 0.49999991 -> 1023), so they are reproduced here.
	float scale = 1.0f, factor = 1.0f;
	for (int tenth = 9; tenth >= 3; tenth--) {
		scale -= 0.1f;
		m.push_back({ "Scale 0." + std::to_string(tenth), "", true, 1, 0.95f, scale, 0.0f });
	}
	for (int tenth = 1; tenth <= 7; tenth++) {
		factor += 0.1f;
		m.push_back({ "Ray step *1." + std::to_string(tenth), "", true, 1, 0.95f, 0.0f, factor });
	}
	return m;
}

void benchmark() {
	printf("Entering benchmark loop...\n\n");
	const std::vector<BenchConfig> matrix = build_matrix();
	const ushort2 original_size = ViewBase::view.dims;
	bool foot_ready = false, foot_tried = false;
	config = 1;
	for (const BenchConfig &c : matrix) {
		if (c.foot_study && !foot_tried) {                   // the studies run on "Foot"; a synthetic volume may stand in
			foot_tried = true;
			foot_ready = benchmark_load_file("Foot") == 0;
			if (!foot_ready && use_synthetic()) {
				printf("(Foot.pvm not available: using the %u^3 synthetic shell for the remaining configurations)\n", synthetic_n);
				foot_ready = install_volume() == 0;
			}
		}
		if (c.foot_study && !foot_ready) break;
		row_names.push_back(c.name);
		printf("%s benchmark\n", c.name.c_str());
		if (!c.dataset.empty() && benchmark_load_file(c.dataset.c_str()) != 0) { config++; continue; }     // VolR.cpp:276-279: a missing file is skipped
		if ((RaycasterBase::raycaster.esl ? 1 : 0) != c.esl) RaycasterBase::toggle_esl();
		RaycasterBase::change_ray_threshold(c.ray_threshold, true);
		if (c.ray_step_factor > 0.0f) {
			RaycasterBase::reset_ray_step();
			RaycasterBase::change_ray_step(RaycasterBase::raycaster.ray_step * c.ray_step_factor, true);
		} else if (c.foot_study) {
			RaycasterBase::reset_ray_step();
		}
		if (c.viewport_scale > 0.0f) {
			ViewBase::set_viewport_dims(original_size, c.viewport_scale);
			printf("Resolution: %dx%d\n", ViewBase::view.dims.x, ViewBase::view.dims.y);
		} else if (ViewBase::view.dims.x != original_size.x || ViewBase::view.dims.y != original_size.y) {
			ViewBase::set_viewport_dims(original_size);
		}
		benchmark_config_loop();
	}
	RaycasterBase::reset_ray_step();
	ViewBase::set_viewport_dims(original_size);
	config--;
}

// VolR.cpp:200-223: the summary table — which statistics a row shows is a property of the row
void print_profiler() {
	printf("\nSummary profiler report:\n");
	for (int r = 0; r < PROFILER_RENDERERS; r++) printf(" Rend.%2i: %s\n", r, renderers[r]->get_name());
	printf("%15s,%8s,", "Configuration", "Value");
	for (int r = 0; r < PROFILER_RENDERERS; r++) printf(" Rend.%2i%s", r, r != PROFILER_RENDERERS - 1 ? "," : "");
	printf("\n");
	typedef void (*Stat)(FILE *, int);
	static const Stat interactive_stats[] = { Profiler::print_samples, Profiler::print_avg, Profiler::print_max };
	static const Stat config_stats[] = { Profiler::print_avg };
	for (int row = 0; row <= config && row < (int) row_names.size(); row++) {
		const Stat *stats = row == 0 ? interactive_stats : config_stats;
		const int count = row == 0 ? 3 : 1;
		for (int k = 0; k < count; k++) { printf("%15s,", row_names[row].c_str()); stats[k](stdout, row); }
	}
}

// premultiplied RGBA8, y-up (RaycasterBase.h:44-50) -> binary PPM over a black background, top row first
int write_ppm(const char *path) {
	FILE *f = fopen(path, "wb");
	if (f == NULL) return 1;
	const int w = ViewBase::view.dims.x, h = ViewBase::view.dims.y;
	fprintf(f, "P6\n%d %d\n255\n", w, h);
	for (int y = h - 1; y >= 0; y--)
		for (int x = 0; x < w; x++) {
			const uchar4 c = frame[(size_t) y * w + x];
			const unsigned char rgb[3] = { c.x, c.y, c.z };
			fwrite(rgb, 1, 3, f);
		}
	fclose(f);
	return 0;
}

}  // namespace

int main(int argc, char **argv) {
	std::string file_name, out_ppm;
	bool benchmark_mode = false, persp = false, have_pose = false;
	float pose[4] = { 120, 0, 200, 3 };       // the reference's interactive start pose (VolR.cpp:436)
	ViewBase::reset();
	for (int i = 1; i < argc; i++) {
		const char *arg = argv[i];
		auto need = [&](int n) { if (i + n >= argc) { printf("%s error: Not enough parameters. Use -h to help.\n", arg); return false; } return true; };
		if (strcmp(arg, "-h") == 0) { print_usage(); return EXIT_SUCCESS; }
		else if (strcmp(arg, "-f") == 0) { if (need(1)) file_name = argv[++i]; }
		else if (strcmp(arg, "-raw") == 0) {
			if (!need(3)) continue;
			const unsigned w = atoi(argv[i + 1]), h = atoi(argv[i + 2]), d = atoi(argv[i + 3]);
			i += 3;
			unsigned c = 1;
			if (i + 1 < argc && argv[i + 1][0] != '-') c = atoi(argv[++i]);
			ModelBase::set_raw_dims(w, h, d, c);
		}
		else if (strcmp(arg, "-synthetic") == 0) { if (need(1)) synthetic_n = atoi(argv[++i]); }
		else if (strcmp(arg, "-dir") == 0) { if (need(1)) dataset_dir = argv[++i]; }
		else if (strcmp(arg, "-r") == 0) {
			if (!need(1)) continue;
			const int r = atoi(argv[++i]);
			if (r < 0 || r >= PROFILER_RENDERERS) { printf("%s error: Wrong parameters. Use -h to help.\n", arg); continue; }
			renderer_id = r;
		}
		else if (strcmp(arg, "-s") == 0) {
			if (!need(2)) continue;
			const int w = atoi(argv[++i]), h = atoi(argv[++i]);
			if (w < 128 || h < 128 || w > 2048 || h > 2048) { printf("%s error: Wrong parameters. Use -h to help.\n", arg); continue; }   // VolR.cpp:392
			ViewBase::set_viewport_dims(make_ushort2((unsigned short) w, (unsigned short) h));
		}
		else if (strcmp(arg, "-d") == 0) { if (need(1)) device = atoi(argv[++i]); }
		else if (strcmp(arg, "-devices") == 0) {
			if (!need(1)) continue;
			for (const char *q = argv[++i]; *q; ) { device_list.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
		}
		else if (strcmp(arg, "-bg") == 0 || strcmp(arg, "-b") == 0) benchmark_mode = true;
		else if (strcmp(arg, "-persp") == 0) persp = true;
		else if (strcmp(arg, "-pose") == 0) { if (need(4)) { for (int k = 0; k < 4; k++) pose[k] = (float) atof(argv[++i]); have_pose = true; } }
		else if (strcmp(arg, "-o") == 0) { if (need(1)) out_ppm = argv[++i]; }
		else printf("Warning: unknown argument: %s\n", arg);
	}

	// VolR.cpp:412-417
	bool loaded = !file_name.empty() && ModelBase::load_model(file_name.c_str()) == 0;
	if (!loaded) loaded = use_synthetic();
	if (!loaded && !benchmark_mode) { printf("Warning: no volume data loaded (use -f or -synthetic).\n"); return EXIT_FAILURE; }
	RaycasterBase::set_view(ViewBase::view);
	RaycasterBase::reset_transfer_fn();
	if (loaded) RaycasterBase::set_volume(ModelBase::volume);
	Profiler::init();

	printf("Initializing renderers 0 - %d...\n", PROFILER_RENDERERS - 1);
	if (!device_list.empty()) {
		renderers[0] = new HipRenderer(RaycasterBase::raycaster, device_list.data(), (int) device_list.size(), VR_SAMPLE_NEAREST);
		renderers[1] = new HipRenderer(RaycasterBase::raycaster, device_list.data(), (int) device_list.size(), VR_SAMPLE_TRILINEAR);
		if (renderers[1]->ok()) printf("Frame split over %d device(s), bands gathered by %s\n", (int) device_list.size(), vr_hip_multi_transport(renderers[1]->multi()));
	} else {
		renderers[0] = new HipRenderer(RaycasterBase::raycaster, device, VR_SAMPLE_NEAREST);
		renderers[1] = new HipRenderer(RaycasterBase::raycaster, device, VR_SAMPLE_TRILINEAR);
	}
	for (int i = 0; i < PROFILER_RENDERERS; i++)
		if (!renderers[i]->ok()) { printf("Error: %s\n", renderers[i]->last_error()); return EXIT_FAILURE; }

	int rc = EXIT_SUCCESS;
	if (benchmark_mode) {
		benchmark();
		print_profiler();
	} else {
		if (persp) ViewBase::toggle_perspective(0);
		ViewBase::set_camera_position(make_float3(pose[0], pose[1], pose[2]), have_pose ? pose[3] : 3.0f);
		draw_volume();
		printf("%s: %dx%d frame in %.2f ms\n", renderers[renderer_id]->get_name(), ViewBase::view.dims.x, ViewBase::view.dims.y, Profiler::time_ms);
		if (!out_ppm.empty() && write_ppm(out_ppm.c_str()) != 0) { printf("Error: cannot write %s\n", out_ppm.c_str()); rc = EXIT_FAILURE; }
	}
	for (int i = PROFILER_RENDERERS - 1; i >= 0; i--) delete renderers[i];      // reverse order like VolR.cpp:328-329
	return rc;
}
