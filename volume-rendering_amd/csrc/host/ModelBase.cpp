// ModelBase.cpp — volume file decoding and the model manager, restated from the reference's behaviour
// (VolumeRendering/ModelBase.cpp, VolumeRendering/ddsbase.cpp); see ModelBase.h.
#include "ModelBase.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace volr {

namespace {

// MSB-first bit reader over a byte string; reads past the end return zeros (ddsbase.cpp pads the stream with zero words
// and returns 0 once the cache is exhausted, DDS_loadbits / DDS_readbits)
class BitReader {
	public:
		BitReader(const uint8_t *p, size_t n) : p_(p), n_(n), pos_(0) {}
		unsigned read(unsigned bits) {
			unsigned v = 0;
			for (unsigned i = 0; i < bits; i++, pos_++) {
				const size_t byte = pos_ >> 3;
				const unsigned bit = byte < n_ ? (p_[byte] >> (7 - (pos_ & 7))) & 1u : 0u;
				v = (v << 1) | bit;
			}
			return v;
		}
	private:
		const uint8_t *p_;
		size_t n_, pos_;
};

// Undoes the encoder's byte de-interleaving (ddsbase.cpp:122-184 with restore = TRUE): the stream stores, per block of
// skip*block bytes (or the whole stream when block == 0), all bytes of lane 0, then lane 1, ...
void reinterleave(std::vector<uint8_t> &data, unsigned skip, unsigned block) {
	if (skip <= 1 || data.empty())
		return;
	const size_t bytes = data.size();
	const size_t span = block == 0 ? bytes : (size_t) skip * block;
	std::vector<uint8_t> tmp(span < bytes ? span : bytes);
	for (size_t base = 0; base < bytes; base += span) {
		const size_t len = bytes - base < span ? bytes - base : span;
		const uint8_t *src = &data[base];
		size_t s = 0;
		for (unsigned lane = 0; lane < skip; lane++)
			for (size_t j = lane; j < len; j += skip)
				tmp[j] = src[s++];
		memcpy(&data[base], tmp.data(), len);
	}
}

bool read_whole(FILE *f, std::vector<uint8_t> *out) {
	out->clear();
	uint8_t buf[1 << 16];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, f)) > 0)
		out->insert(out->end(), buf, buf + n);
	return !out->empty();
}

}  // namespace

// ddsbase.cpp:187-245 DDS_decode.  Stream: 2 bits skip-1, 16 bits strip-1, then runs {7 bits count (0 = end), 3 bits width
// code w -> w ? w + 1 : 0 bits, count deltas of that width biased by half the range}.  Each delta updates a running value:
// plain DPCM for the first `strip` samples (or strip == 1), afterwards DPCM of the difference to the sample one strip back
// (2-D prediction).  Values wrap modulo 256.
bool dds_decode(const uint8_t *chunk, size_t size, unsigned block, std::vector<uint8_t> *out) {
	BitReader in(chunk, size);
	const unsigned skip = in.read(2) + 1;
	const unsigned strip = in.read(16) + 1;
	out->clear();
	int act = 0;
	for (;;) {
		const unsigned run = in.read(7);
		if (run == 0)
			break;
		const unsigned code = in.read(3);
		const unsigned bits = code >= 1 ? code + 1 : code;
		const int bias = (1 << bits) / 2;
		for (unsigned i = 0; i < run; i++) {
			const size_t cnt = out->size();
			const int delta = (int) in.read(bits) - bias;
			if (strip == 1 || cnt <= strip)
				act += delta;
			else
				act += (int) (*out)[cnt - strip] - (int) (*out)[cnt - strip - 1] + delta;
			while (act < 0) act += 256;
			while (act > 255) act -= 256;
			out->push_back((uint8_t) act);
		}
	}
	reinterleave(*out, skip, block);
	return true;
}

bool read_raw_file(const char *file_name, std::vector<uint8_t> *out) {
	FILE *f = fopen(file_name, "rb");
	if (f == NULL)
		return false;
	const bool ok = read_whole(f, out);
	fclose(f);
	return ok;
}

// ddsbase.cpp:298-342 readDDSfile + :345-435 readPVMvolume
bool read_pvm_volume(const char *file_name, PvmVolume *v) {
	std::vector<uint8_t> file, data;
	if (!read_raw_file(file_name, &file))
		return false;
	static const char kV3d[] = "DDS v3d\n", kV3e[] = "DDS v3e\n";
	if (file.size() >= 8 && memcmp(file.data(), kV3d, 8) == 0)
		dds_decode(file.data() + 8, file.size() - 8, 0, &data);
	else if (file.size() >= 8 && memcmp(file.data(), kV3e, 8) == 0)
		dds_decode(file.data() + 8, file.size() - 8, 1u << 24, &data);
	else
		data.swap(file);                            // not a DDS stream: an uncompressed PVM file
	if (data.size() < 5)
		return false;
	data.push_back(0);                              // the header is parsed as text
	const char *text = (const char *) data.data();
	const char *end = text + data.size() - 1;
	const char *p;
	int version = 1;
	v->scale[0] = v->scale[1] = v->scale[2] = 1.0f;
	if (strncmp(text, "PVM\n", 4) == 0) {
		p = text + 4;
		while (p < end && *p == '#') {              // comment lines; a '#' line without a newline ends the header scan at the buffer end
			while (p < end && *p != '\n') p++;
			if (p < end) p++;
		}
		if (sscanf(p, "%u %u %u\n", &v->width, &v->height, &v->depth) != 3)
			return false;
	} else {
		if (strncmp(text, "PVM2\n", 5) == 0) version = 2;
		else if (strncmp(text, "PVM3\n", 5) == 0) version = 3;
		else return false;
		p = text + 5;
		if (sscanf(p, "%u %u %u\n%g %g %g\n", &v->width, &v->height, &v->depth, &v->scale[0], &v->scale[1], &v->scale[2]) != 6)
			return false;
		if (v->scale[0] <= 0.0f || v->scale[1] <= 0.0f || v->scale[2] <= 0.0f)
			return false;
		p = strchr(p, '\n');
		if (p == NULL) return false;
		p++;
	}
	if (v->width < 1 || v->height < 1 || v->depth < 1 || v->width > 65535u || v->height > 65535u || v->depth > 65535u)
		return false;                               // Model::dims is ushort3 (ModelBase.h:13); also keeps the byte count below 2^48 * components
	p = strchr(p, '\n');
	if (p == NULL) return false;
	p++;
	if (sscanf(p, "%u\n", &v->components) != 1 || v->components < 1 || v->components > 2)
		return false;                               // 8- or 16-bit samples only (ddsbase.cpp:475-558 quantises 2 -> 1)
	p = strchr(p, '\n');
	if (p == NULL) return false;
	p++;
	const size_t voxel_bytes = (size_t) v->width * v->height * v->depth * v->components;
	if ((size_t) (end - p) < voxel_bytes)
		return false;
	const char *q = p + voxel_bytes;
	std::string *strings[4] = { &v->description, &v->courtesy, &v->parameter, &v->comment };
	for (int i = 0; i < 4; i++) {
		strings[i]->clear();
		if (version == 3) {                         // four zero-terminated strings follow the voxels
			if (q > end) return false;
			const size_t len = strnlen(q, (size_t) (end - q));
			strings[i]->assign(q, len);
			q += len + 1;
		}
	}
	if (q != end)                                   // the reference insists that the sizes add up exactly
		return false;
	v->voxels.assign((const uint8_t *) p, (const uint8_t *) p + voxel_bytes);
	return true;
}

// ddsbase.cpp:437-558 quantize.  The non-linear map spends the 256 output levels where the volume has structure: every
// 16-bit value accumulates sqrt(|gradient|) over the voxels that carry it, the cube root of that is clipped iteratively
// to 1/256 of its integral, and the running integral scaled to 0..255 is the lookup table.
std::vector<uint8_t> quantize_16_to_8(const uint8_t *data, unsigned width, unsigned height, unsigned depth, bool linear) {
	const size_t n = (size_t) width * height * depth;
	std::vector<uint16_t> s(n);
	int vmin = 65535, vmax = 0;
	for (size_t i = 0; i < n; i++) {
		const int v = 256 * data[2 * i] + data[2 * i + 1];
		s[i] = (uint16_t) v;
		if (v < vmin) vmin = v;
		if (v > vmax) vmax = v;
	}
	std::vector<double> err(65536, 0.0);
	if (linear) {
		for (int i = 0; i < 65536; i++) err[i] = 255 * (double) i / vmax;
	} else {
		auto at = [&](unsigned i, unsigned j, unsigned k) { return (int) s[i + ((size_t) j + (size_t) k * height) * width]; };
		// central differences inside, one-sided at the faces, 0 across a single-voxel axis
		auto diff = [](int lo, int mid, int hi, bool has_lo, bool has_hi) {
			if (has_lo) return has_hi ? (hi - lo) / 2.0 : (double) (mid - lo);
			return has_hi ? (double) (hi - mid) : 0.0;
		};
		for (unsigned k = 0; k < depth; k++)
			for (unsigned j = 0; j < height; j++)
				for (unsigned i = 0; i < width; i++) {
					const int c = at(i, j, k);
					const bool xl = i > 0, xh = i < width - 1, yl = j > 0, yh = j < height - 1, zl = k > 0, zh = k < depth - 1;
					const double gx = diff(xl ? at(i - 1, j, k) : 0, c, xh ? at(i + 1, j, k) : 0, xl, xh);
					const double gy = diff(yl ? at(i, j - 1, k) : 0, c, yh ? at(i, j + 1, k) : 0, yl, yh);
					const double gz = diff(zl ? at(i, j, k - 1) : 0, c, zh ? at(i, j, k + 1) : 0, zl, zh);
					err[c] += sqrt(sqrt(gx * gx + gy * gy + gz * gz));
				}
		for (int i = 0; i < 65536; i++) err[i] = pow(err[i], 1.0 / 3);
		err[vmin] = err[vmax] = 0.0;
		for (int pass = 0; pass < 256; pass++) {
			double eint = 0.0;
			for (int i = 0; i < 65536; i++) eint += err[i];
			bool done = true;
			for (int i = 0; i < 65536; i++)
				if (err[i] > eint / 256) { err[i] = eint / 256; done = false; }
			if (done) break;
		}
		for (int i = 1; i < 65536; i++) err[i] += err[i - 1];
		if (err[65535] > 0.0f)
			for (int i = 0; i < 65536; i++) err[i] *= 255.0f / err[65535];
	}
	std::vector<uint8_t> out(n);
	for (size_t i = 0; i < n; i++)
		out[i] = (uint8_t) (int) (err[s[i]] + 0.5);
	return out;
}

// ---- ModelBase --------------------------------------------------------------------------------------------------------

Model ModelBase::volume = { NULL, 0, { 0, 0, 0 }, { -1.0f, -1.0f, -1.0f } };
float ModelBase::histogram[256];
char ModelBase::file_name[256] = "";
unsigned ModelBase::raw_dims[4] = { 0, 0, 0, 1 };

void ModelBase::set_raw_dims(unsigned width, unsigned height, unsigned depth, unsigned components) {
	raw_dims[0] = width; raw_dims[1] = height; raw_dims[2] = depth; raw_dims[3] = components;
}

// ModelBase.cpp:19-33: fourth root of the bin counts, normalised to the largest bin
void ModelBase::compute_histogram() {
	unsigned int counts[256] = { 0 };
	for (unsigned int i = 0; i < volume.size; i++)
		counts[volume.data[i]]++;
	float max_value = 0;
	for (int i = 0; i < 256; i++) {
		histogram[i] = sqrtf(sqrtf((float) counts[i]));
		if (histogram[i] > max_value) max_value = histogram[i];
	}
	for (int i = 0; i < 256; i++)
		histogram[i] = histogram[i] / max_value;
}

// ModelBase.cpp:35-109
int ModelBase::load_model(const char *name) {
	const char *dot = strrchr(name, '.');
	const bool is_raw = dot != NULL && strcmp(dot, ".raw") == 0, is_pvm = dot != NULL && strcmp(dot, ".pvm") == 0;
	if (!is_raw && !is_pvm)
		return 1;                                   // unsupported extension
	unsigned width, height, depth, components = 1;
	std::vector<uint8_t> bytes;
	if (is_pvm) {
		PvmVolume v;
		if (!read_pvm_volume(name, &v))
			return 1;
		width = v.width; height = v.height; depth = v.depth; components = v.components;
		bytes.swap(v.voxels);
	} else {
		if (!read_raw_file(name, &bytes))
			return 1;
		width = raw_dims[0]; height = raw_dims[1]; depth = raw_dims[2]; components = raw_dims[3];
		if ((size_t) width * height * depth * components != bytes.size())
			return 1;                               // "Incorrect RAW file volume parameters"
	}
	if (components > 2)
		return 1;
	if (width < 1 || height < 1 || depth < 1 || width > 65535u || height > 65535u || depth > 65535u)
		return 1;                                   // checked BEFORE the quantiser walks width * height * depth samples
	if (components == 2)
		bytes = quantize_16_to_8(bytes.data(), width, height, depth);
	unsigned char *copy = (unsigned char *) malloc(bytes.size());
	if (copy == NULL)
		return 1;
	memcpy(copy, bytes.data(), bytes.size());
	if (volume.data != NULL)
		free(volume.data);
	volume.dims = make_ushort3((unsigned short) width, (unsigned short) height, (unsigned short) depth);
	volume.size = (unsigned int) ((size_t) width * height * depth);
	volume.data = copy;
	strncpy(file_name, name, sizeof file_name - 1);
	file_name[sizeof file_name - 1] = '\0';
	compute_histogram();
	return 0;
}

}  // namespace volr
