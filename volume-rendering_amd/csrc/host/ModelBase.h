// ModelBase.h — host mirror of the reference's model manager (VolumeRendering/ModelBase.h:26-33, ModelBase.cpp) and of
// the volume file formats it reads through the third-party V^3 codec (VolumeRendering/ddsbase.{h,cpp}, S. Roettger):
// PVM containers ("PVM\n" / "PVM2\n" / "PVM3\n"), optionally wrapped in a DDS differential bit stream ("DDS v3d\n",
// "DDS v3e\n"), plain RAW files, and the 16 -> 8 bit quantisation.  SURVEY §8 row f1.  Everything here is serial
// integer / byte work on the host; the decoded voxels then go through Renderer::set_volume.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "Renderer.h"

namespace volr {

// ---- codec layer (what ddsbase.cpp provides to the reference) ----------------------------------------------------

// Decodes a DDS differential data stream (the bytes AFTER the 8-byte identifier).  interleave_block = 0 for "DDS v3d",
// 1 << 24 for "DDS v3e".  Returns false on a malformed stream.
bool dds_decode(const uint8_t *chunk, size_t size, unsigned interleave_block, std::vector<uint8_t> *out);

struct PvmVolume {
	std::vector<uint8_t> voxels;                    // width*height*depth*components bytes (16-bit samples big-endian)
	unsigned width = 0, height = 0, depth = 0, components = 0;
	float scale[3] = { 1.0f, 1.0f, 1.0f };
	std::string description, courtesy, parameter, comment;   // PVM3 only
};

// readPVMvolume: DDS-compressed or plain PVM file.  Returns false if the file is missing or not a PVM volume.
bool read_pvm_volume(const char *file_name, PvmVolume *out);
bool read_raw_file(const char *file_name, std::vector<uint8_t> *out);
// quantize(): 16-bit big-endian samples -> 8 bit, non-linear (gradient-weighted error integral) or linear mapping
std::vector<uint8_t> quantize_16_to_8(const uint8_t *data, unsigned width, unsigned height, unsigned depth, bool linear = false);

// ---- ModelBase (VolumeRendering/ModelBase.h:26-33) -----------------------------------------------------------------

class ModelBase {
	public:
		static Model volume;
		static float histogram[256];
		static char file_name[256];
		// .pvm: self-describing.  .raw: the reference asks for the dimensions on stdin (ModelBase.cpp:78-88); here they
		// come from set_raw_dims() beforehand (a headless library cannot prompt).  Returns 0 / 1 like the reference.
		static int load_model(const char *file_name);
		static void set_raw_dims(unsigned width, unsigned height, unsigned depth, unsigned components = 1);
	private:
		static void compute_histogram();
		static unsigned raw_dims[4];
};

}  // namespace volr
