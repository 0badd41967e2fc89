#include "blend_reader.h"

#include <cctype>
#include <cstdio>

namespace acr
{
namespace blend
{
	namespace
	{
		uint32_t rd32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
		uint64_t rd64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
		uint16_t rd16(const uint8_t *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

		// block codes are two capitals + two NULs ("OB\0\0") or four capitals/digits ("DATA", "DNA1")
		bool plausibleCode(const uint8_t *p)
		{
			auto up = [](uint8_t c) { return (c >= 'A' && c <= 'Z'); };
			auto upd = [&](uint8_t c) { return up(c) || (c >= '0' && c <= '9'); };
			if (up(p[0]) && up(p[1]) && p[2] == 0 && p[3] == 0) return true;
			return upd(p[0]) && upd(p[1]) && upd(p[2]) && upd(p[3]);
		}

		struct DnaTables
		{
			std::vector<std::string> names, types;
			std::vector<uint16_t> lengths;
			struct RawStruct { uint16_t type; std::vector<std::pair<uint16_t, uint16_t>> fields; };
			std::vector<RawStruct> structs;
		};

		// Strict SDNA parse: every section tag present, every index in range, the
		// block consumed to within alignment slack, and every struct's fields
		// adding up to its declared length.  Strict on purpose: the repair below
		// uses "parses cleanly" as its test.
		bool parseTables(const uint8_t *p, size_t n, DnaTables &t)
		{
			size_t at = 0;
			auto tag = [&](const char *s) {
				if (at + 4 > n || std::memcmp(p + at, s, 4) != 0) return false;
				at += 4;
				return true;
			};
			auto align4 = [&]() { at = (at + 3) & ~size_t(3); };
			auto strings = [&](std::vector<std::string> &out) {
				if (at + 4 > n) return false;
				const uint32_t count = rd32(p + at);
				at += 4;
				if (count > 100000) return false;
				for (uint32_t i = 0; i < count; i++) {
					const size_t b = at;
					while (at < n && p[at] != 0) at++;
					if (at >= n) return false;
					out.emplace_back(reinterpret_cast<const char *>(p + b), at - b);
					at++;
				}
				return true;
			};
			if (!tag("SDNA") || !tag("NAME") || !strings(t.names)) return false;
			align4();
			if (!tag("TYPE") || !strings(t.types)) return false;
			align4();
			if (!tag("TLEN")) return false;
			if (at + 2 * t.types.size() > n) return false;
			for (size_t i = 0; i < t.types.size(); i++) t.lengths.push_back(rd16(p + at + 2 * i));
			at += 2 * t.types.size();
			align4();
			if (!tag("STRC") || at + 4 > n) return false;
			const uint32_t sc = rd32(p + at);
			at += 4;
			if (sc > 100000) return false;
			for (uint32_t i = 0; i < sc; i++) {
				if (at + 4 > n) return false;
				DnaTables::RawStruct rs;
				rs.type = rd16(p + at);
				const uint16_t nf = rd16(p + at + 2);
				at += 4;
				if (rs.type >= t.types.size() || at + 4 * size_t(nf) > n) return false;
				for (uint16_t f = 0; f < nf; f++) {
					const uint16_t ft = rd16(p + at), fn = rd16(p + at + 2);
					at += 4;
					if (ft >= t.types.size() || fn >= t.names.size()) return false;
					rs.fields.emplace_back(ft, fn);
				}
				t.structs.push_back(rs);
			}
			return n - at < 4;
		}

		// size and shape of one field from its DNA name: "*next", "obmat[4][4]", "(*func)()"
		void fieldShape(const std::string &raw, size_t typeLen, size_t ptrSize, Field &f)
		{
			f.pointer = !raw.empty() && (raw[0] == '*' || raw[0] == '(');
			size_t b = 0;
			while (b < raw.size() && (raw[b] == '*' || raw[b] == '(')) b++;
			size_t e = b;
			while (e < raw.size() && (std::isalnum(uint8_t(raw[e])) || raw[e] == '_')) e++;
			f.name = raw.substr(b, e - b);
			f.elems = 1;
			for (size_t k = e; k < raw.size(); k++) {
				if (raw[k] == '[') {
					f.elems *= uint32_t(std::strtoul(raw.c_str() + k + 1, nullptr, 10));
				}
			}
			f.size = (f.pointer ? ptrSize : typeLen) * f.elems;
		}
	} // namespace

	bool File::parseDna(const uint8_t *p, size_t n, std::string &error)
	{
		DnaTables t;
		if (!parseTables(p, n, t)) {
			error = "SDNA does not parse";
			return false;
		}
		structs.clear();
		for (const auto &rs : t.structs) {
			Struct s;
			s.name = t.types[rs.type];
			s.size = t.lengths[rs.type];
			size_t off = 0;
			for (const auto &fp : rs.fields) {
				Field f;
				f.type = t.types[fp.first];
				fieldShape(t.names[fp.second], t.lengths[fp.first], 8, f);
				f.offset = off;
				off += f.size;
				s.fields.push_back(f);
			}
			if (off != s.size) {
				error = "SDNA struct " + s.name + ": fields do not add up to its length";
				return false;
			}
			structs.push_back(s);
		}
		return true;
	}

	bool File::open(const char *path, std::string &error)
	{
		FILE *fp = std::fopen(path, "rb");
		if (!fp) {
			error = std::string("cannot open ") + path;
			return false;
		}
		std::fseek(fp, 0, SEEK_END);
		const long len = std::ftell(fp);
		std::fseek(fp, 0, SEEK_SET);
		image.resize(len > 0 ? size_t(len) : 0);
		const size_t got = image.empty() ? 0 : std::fread(image.data(), 1, image.size(), fp);
		std::fclose(fp);
		if (got != image.size() || image.size() < 12 + 24) {
			error = "short read";
			return false;
		}
		if (std::memcmp(image.data(), "BLENDER", 7) != 0) {
			error = "not an uncompressed .blend file";
			return false;
		}
		if (image[7] != '-' || image[8] != 'v') {
			error = "only 64-bit little-endian .blend files are supported";
			return false;
		}
		ver = (image[9] - '0') * 100 + (image[10] - '0') * 10 + (image[11] - '0');

		blockList.clear();
		byPointer.clear();
		repairCount = 0;
		size_t at = 12;
		bool sawDna = false;
		while (at + 24 <= image.size()) {
			if (!plausibleCode(image.data() + at)) {
				// a payload that lost bytes: the header sits a little earlier than computed
				bool found = false;
				for (size_t back = 1; back <= 8 && back <= at; back++) {
					if (plausibleCode(image.data() + at - back)) {
						at -= back;
						repairCount++;
						found = true;
						break;
					}
				}
				if (!found) {
					error = "lost track of the block chain";
					return false;
				}
			}
			Block b;
			std::memcpy(b.code, image.data() + at, 4);
			b.code[4] = 0;
			b.size = rd32(image.data() + at + 4);
			b.oldPtr = rd64(image.data() + at + 8);
			b.sdnaIndex = rd32(image.data() + at + 16);
			b.count = rd32(image.data() + at + 20);
			b.offset = at + 24;
			if (std::strcmp(b.code, "ENDB") == 0) break;
			if (std::strcmp(b.code, "DNA1") == 0) {
				// the payload ends where the ENDB header starts; if that is short of the
				// declared size, a byte went missing inside: try giving back a 0x0D in
				// front of each 0x0A until the tables parse cleanly
				size_t avail = image.size() - b.offset;
				for (size_t k = b.offset; k + 4 <= image.size(); k++) {
					if (std::memcmp(image.data() + k, "ENDB", 4) == 0 && k + 24 >= image.size() - 8) { avail = k - b.offset; }
				}
				const uint8_t *p = image.data() + b.offset;
				std::string dnaError;
				if (avail >= b.size) {
					if (!parseDna(p, b.size, dnaError)) { error = dnaError; return false; }
				} else if (b.size - avail == 1) {
					bool ok = false;
					std::vector<uint8_t> fixed(b.size);
					for (size_t k = 0; k < avail && !ok; k++) {
						if (p[k] != 0x0A) continue;
						std::memcpy(fixed.data(), p, k);
						fixed[k] = 0x0D;
						std::memcpy(fixed.data() + k + 1, p + k, avail - k);
						ok = parseDna(fixed.data(), fixed.size(), dnaError);
					}
					if (!ok) { error = "damaged DNA1 block could not be repaired"; return false; }
					repairCount++;
				} else {
					error = "DNA1 block is short by more than one byte";
					return false;
				}
				sawDna = true;
				blockList.push_back(b);
				break;
			}
			blockList.push_back(b);
			at = b.offset + b.size;
		}
		if (!sawDna) {
			error = "no DNA1 block";
			return false;
		}
		for (size_t i = 0; i < blockList.size(); i++)
			if (blockList[i].oldPtr) byPointer.emplace(blockList[i].oldPtr, i);
		return true;
	}

	const Struct *File::structByName(const std::string &name) const
	{
		for (const Struct &s : structs)
			if (s.name == name) return &s;
		return nullptr;
	}

	const Block *File::blockAt(uint64_t ptr) const
	{
		if (!ptr) return nullptr;
		auto it = byPointer.find(ptr);
		return it == byPointer.end() ? nullptr : &blockList[it->second];
	}

	bool File::resolve(const Struct &s, const std::string &path, Field &out) const
	{
		const size_t dot = path.find('.');
		const std::string head = dot == std::string::npos ? path : path.substr(0, dot);
		const Field *f = s.find(head);
		if (!f) return false;
		if (dot == std::string::npos) {
			out = *f;
			return true;
		}
		if (f->pointer) return false;
		const Struct *inner = structByName(f->type);
		if (!inner) return false;
		Field sub;
		if (!resolve(*inner, path.substr(dot + 1), sub)) return false;
		sub.offset += f->offset;
		out = sub;
		return true;
	}

	std::string File::readString(const uint8_t *base, const Struct &s, const std::string &path) const
	{
		Field f;
		if (!resolve(s, path, f)) return std::string();
		const char *p = reinterpret_cast<const char *>(base + f.offset);
		size_t n = 0;
		while (n < f.size && p[n]) n++;
		return std::string(p, n);
	}
} // namespace blend
} // namespace acr
