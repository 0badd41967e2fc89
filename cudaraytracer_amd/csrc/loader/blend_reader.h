// Minimal reader for Blender .blend files: file-block walk, SDNA (struct DNA)
// parse, field access by name.  Handles what the reference's shipped scenes
// need: uncompressed, little-endian, 64-bit-pointer files.
//
// The reference does not read .blend itself; it hands the path to assimp
// (src/scene/scene.cu:30-35), which is not part of its tree.  This reader is
// the first half of the replacement; loader/blend_import.cpp is the second.
#ifndef ACR_LOADER_BLEND_READER_H
#define ACR_LOADER_BLEND_READER_H

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace acr
{
namespace blend
{
	struct Block
	{
		char code[5];
		uint32_t size;      // declared byte size of the payload
		uint64_t oldPtr;    // address the block had in Blender's memory: the key pointers refer to
		uint32_t sdnaIndex; // struct of the payload
		uint32_t count;     // number of consecutive structs
		size_t offset;      // payload position in the (repaired) file image
	};

	struct Field
	{
		std::string type;  // DNA type name
		std::string name;  // bare name, without '*', '[n]'
		size_t offset = 0;
		size_t size = 0;   // whole field, arrays included
		bool pointer = false;
		uint32_t elems = 1; // product of array dimensions
	};

	struct Struct
	{
		std::string name;
		size_t size = 0;
		std::vector<Field> fields;
		const Field *find(const std::string &n) const
		{
			for (const Field &f : fields)
				if (f.name == n) return &f;
			return nullptr;
		}
	};

	class File
	{
	public:
		// Reads and indexes the file.  Repairs the two kinds of damage a CRLF->LF
		// text filter leaves in a binary file (a 0x0D dropped in front of a 0x0A):
		// a block header that no longer sits where the previous block's size says,
		// and a DNA1 block that is shorter than declared.
		bool open(const char *path, std::string &error);

		int version() const { return ver; }
		const std::vector<Block> &blocks() const { return blockList; }
		const Struct *structByName(const std::string &name) const;
		const Struct *structOf(const Block &b) const { return b.sdnaIndex < structs.size() ? &structs[b.sdnaIndex] : nullptr; }
		// block whose payload starts at Blender address `ptr` (nullptr for 0 / unknown)
		const Block *blockAt(uint64_t ptr) const;
		const uint8_t *payload(const Block &b) const { return image.data() + b.offset; }
		size_t payloadAvailable(const Block &b) const { return b.offset <= image.size() ? image.size() - b.offset : 0; }
		int repairs() const { return repairCount; }

		// Resolves a dotted field path ("id.name", "r.xsch") inside struct `s`;
		// returns the leaf field with its offset relative to the start of `s`.
		bool resolve(const Struct &s, const std::string &path, Field &out) const;

		template <typename T>
		T read(const uint8_t *base, const Struct &s, const std::string &path, uint32_t elem = 0) const
		{
			Field f;
			T v{};
			if (resolve(s, path, f) && elem < f.elems && sizeof(T) * f.elems <= f.size)
				std::memcpy(&v, base + f.offset + sizeof(T) * elem, sizeof(T));
			return v;
		}
		uint64_t readPointer(const uint8_t *base, const Struct &s, const std::string &path) const { return read<uint64_t>(base, s, path); }
		std::string readString(const uint8_t *base, const Struct &s, const std::string &path) const;

	private:
		bool parseDna(const uint8_t *p, size_t n, std::string &error);

		std::vector<uint8_t> image;
		std::vector<Block> blockList;
		std::map<uint64_t, size_t> byPointer;
		std::vector<Struct> structs;
		int ver = 0;
		int repairCount = 0;
	};
} // namespace blend
} // namespace acr

#endif
