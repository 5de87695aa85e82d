// qk_hdf5_mini.hpp — the part of the HDF5 file format the cloudy_cooling_tools tables use, read without libhdf5 (there is none in this image):
// version-0 superblock, 8-byte offsets and lengths, the root group as a version-1 B-tree of symbol-table nodes over a local heap, version-1 object
// headers (continuation blocks followed), simple dataspaces, fixed-point / floating-point datatypes of either byte order, contiguous or compact
// layout, version-1 attributes.  Anything else (chunked or filtered datasets, newer superblocks, nested groups) is reported as an error.
// Written from the format specification ("HDF5 File Format Specification Version 1.1"), which the reference reads through H5Dread / H5Aread
// (src/cooling/CloudyDataReader.cpp:59-199).
#ifndef QK_HDF5_MINI_HPP_
#define QK_HDF5_MINI_HPP_

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace qk
{
namespace h5
{

struct Datatype {
	int cls = -1; // 0 fixed point, 1 floating point
	int size = 0;
	bool bigEndian = false;
	bool isSigned = false;
};

struct Attribute {
	Datatype type;
	std::vector<uint64_t> dims;
	std::vector<unsigned char> raw;
};

struct Dataset {
	Datatype type;
	std::vector<uint64_t> dims;
	std::vector<unsigned char> raw;
	std::map<std::string, Attribute> attrs;

	[[nodiscard]] auto count() const -> size_t
	{
		size_t n = 1;
		for (auto d : dims) {
			n *= static_cast<size_t>(d);
		}
		return n;
	}
};

inline void swapBytes(unsigned char *p, int size)
{
	for (int a = 0, b = size - 1; a < b; ++a, --b) {
		std::swap(p[a], p[b]);
	}
}

// elements of `raw` as native doubles (from 4- or 8-byte floats) or 64-bit integers (from 1/2/4/8-byte fixed point)
inline auto asDoubles(Datatype const &t, std::vector<unsigned char> raw, size_t n) -> std::vector<double>
{
	if (t.cls != 1 || (t.size != 8 && t.size != 4) || raw.size() < n * static_cast<size_t>(t.size)) {
		throw std::runtime_error("hdf5: not a floating-point dataset of 4 or 8 bytes");
	}
	std::vector<double> out(n);
	for (size_t i = 0; i < n; ++i) {
		unsigned char *p = raw.data() + i * t.size;
		if (t.bigEndian) {
			swapBytes(p, t.size);
		}
		if (t.size == 8) {
			std::memcpy(&out[i], p, 8);
		} else {
			float f = 0;
			std::memcpy(&f, p, 4);
			out[i] = f;
		}
	}
	return out;
}
inline auto asIntegers(Datatype const &t, std::vector<unsigned char> raw, size_t n) -> std::vector<int64_t>
{
	if (t.cls != 0 || t.size < 1 || t.size > 8 || raw.size() < n * static_cast<size_t>(t.size)) {
		throw std::runtime_error("hdf5: not a fixed-point value");
	}
	std::vector<int64_t> out(n);
	for (size_t i = 0; i < n; ++i) {
		unsigned char *p = raw.data() + i * t.size;
		if (t.bigEndian) {
			swapBytes(p, t.size);
		}
		uint64_t v = 0;
		std::memcpy(&v, p, static_cast<size_t>(t.size)); // (little-endian host)
		if (t.isSigned && t.size < 8 && ((v >> (8 * t.size - 1)) & 1U) != 0) {
			v |= ~uint64_t(0) << (8 * t.size);
		}
		out[i] = static_cast<int64_t>(v);
	}
	return out;
}

class File
{
      public:
	explicit File(std::string const &path)
	{
		std::ifstream f(path, std::ifstream::binary);
		if (!f.good()) {
			throw std::runtime_error("hdf5: cannot open " + path);
		}
		d_.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
		static const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
		need(0, 96);
		if (std::memcmp(d_.data(), sig, 8) != 0) {
			throw std::runtime_error("hdf5: " + path + " is not an HDF5 file");
		}
		if (d_[8] != 0 || d_[13] != 8 || d_[14] != 8) {
			throw std::runtime_error("hdf5: only version-0 superblocks with 8-byte offsets and lengths are read");
		}
		// root group symbol-table entry at 56: link name offset, object header address, cache type, reserved, scratch (B-tree, heap)
		uint64_t const btree = u64(56 + 24), heap = u64(56 + 32);
		if (u32(56 + 16) != 1) {
			throw std::runtime_error("hdf5: the root group carries no cached symbol table");
		}
		uint64_t const names = heapData(heap);
		walk(btree, names, 0);
	}

	[[nodiscard]] auto has(std::string const &name) const -> bool { return objects_.count(name) != 0; }

	// the dataset /name with its attributes
	[[nodiscard]] auto dataset(std::string const &name) const -> Dataset
	{
		auto it = objects_.find(name);
		if (it == objects_.end()) {
			throw std::runtime_error("hdf5: no object /" + name);
		}
		Dataset ds;
		bool haveLayout = false;
		for (auto const &m : messages(it->second)) {
			unsigned char const *b = d_.data() + m.at;
			switch (m.type) {
			case 0x1:
				ds.dims = dataspace(b, m.size);
				break;
			case 0x3:
				ds.type = datatype(b, m.size);
				break;
			case 0x8: {
				if (m.size < 4 || b[0] != 3) {
					throw std::runtime_error("hdf5: /" + name + ": only version-3 data layout messages are read");
				}
				if (b[1] == 1) { // contiguous
					uint64_t const addr = rd64(b + 2), size = rd64(b + 10);
					need(addr, size);
					ds.raw.assign(d_.begin() + static_cast<std::ptrdiff_t>(addr), d_.begin() + static_cast<std::ptrdiff_t>(addr + size));
				} else if (b[1] == 0) { // compact
					uint64_t const size = static_cast<uint64_t>(b[2]) | (static_cast<uint64_t>(b[3]) << 8);
					within(m, 4 + size, name);
					ds.raw.assign(b + 4, b + 4 + size);
				} else {
					throw std::runtime_error("hdf5: /" + name + " is chunked: not read");
				}
				haveLayout = true;
				break;
			}
			case 0xB:
				throw std::runtime_error("hdf5: /" + name + " has a filter pipeline: not read");
			case 0xC: {
				if (b[0] != 1) {
					throw std::runtime_error("hdf5: /" + name + ": only version-1 attribute messages are read");
				}
				auto pad = [](size_t n) { return (n + 7) / 8 * 8; };
				within(m, 8, name);
				size_t const nsz = rd16(b + 2), dsz = rd16(b + 4), ssz = rd16(b + 6);
				size_t p = 8;
				within(m, p + pad(nsz) + pad(dsz) + pad(ssz), name); // every size below comes from the file: nothing is read beyond the message
				std::string const aname(reinterpret_cast<char const *>(b + p), strnlen(reinterpret_cast<char const *>(b + p), nsz));
				p += pad(nsz);
				Attribute a;
				a.type = datatype(b + p, dsz);
				p += pad(dsz);
				a.dims = dataspace(b + p, ssz);
				p += pad(ssz);
				size_t n = 1;
				for (auto dd : a.dims) {
					if (dd > m.size) { // (an attribute's values live inside its message)
						throw std::runtime_error("hdf5: /" + name + ": attribute " + aname + " is larger than its message");
					}
					n *= static_cast<size_t>(dd);
				}
				if (a.type.size < 0) {
					throw std::runtime_error("hdf5: /" + name + ": attribute " + aname + " has a negative element size");
				}
				within(m, p + n * static_cast<size_t>(a.type.size), name);
				a.raw.assign(b + p, b + p + n * static_cast<size_t>(a.type.size));
				ds.attrs[aname] = a;
				break;
			}
			default:
				break;
			}
		}
		if (!haveLayout || ds.type.cls < 0) {
			throw std::runtime_error("hdf5: /" + name + " is not a dataset");
		}
		return ds;
	}

      private:
	struct Msg {
		int type;
		size_t at, size;
	};
	std::vector<unsigned char> d_;
	std::map<std::string, uint64_t> objects_; // name in the root group -> object header address

	// [0, n) of message m lies inside the message (its extent inside the file was checked when the header was walked)
	static void within(Msg const &m, uint64_t n, std::string const &name)
	{
		if (n > m.size) {
			throw std::runtime_error("hdf5: /" + name + ": a size field points beyond its header message (truncated or malformed file)");
		}
	}
	void need(uint64_t at, uint64_t n) const
	{
		if (at + n > d_.size() || at + n < at) {
			throw std::runtime_error("hdf5: address beyond the end of the file");
		}
	}
	static auto rd16(unsigned char const *p) -> uint32_t { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8); }
	static auto rd32(unsigned char const *p) -> uint32_t { return rd16(p) | (rd16(p + 2) << 16); }
	static auto rd64(unsigned char const *p) -> uint64_t { return static_cast<uint64_t>(rd32(p)) | (static_cast<uint64_t>(rd32(p + 4)) << 32); }
	[[nodiscard]] auto u16(uint64_t at) const -> uint32_t
	{
		need(at, 2);
		return rd16(d_.data() + at);
	}
	[[nodiscard]] auto u32(uint64_t at) const -> uint32_t
	{
		need(at, 4);
		return rd32(d_.data() + at);
	}
	[[nodiscard]] auto u64(uint64_t at) const -> uint64_t
	{
		need(at, 8);
		return rd64(d_.data() + at);
	}
	[[nodiscard]] auto tag(uint64_t at, char const *t) const -> bool
	{
		need(at, 4);
		return std::memcmp(d_.data() + at, t, 4) == 0;
	}
	[[nodiscard]] auto heapData(uint64_t heap) const -> uint64_t
	{
		if (!tag(heap, "HEAP")) {
			throw std::runtime_error("hdf5: local heap expected");
		}
		return u64(heap + 24);
	}
	void walk(uint64_t node, uint64_t names, int depth)
	{
		need(node, 24);
		if (depth > 8 || !tag(node, "TREE") || d_[node + 4] != 0) {
			throw std::runtime_error("hdf5: group B-tree node expected");
		}
		int const level = d_[node + 5];
		uint32_t const used = u16(node + 6);
		uint64_t p = node + 24; // signature, type, level, entries, two sibling addresses
		for (uint32_t n = 0; n < used; ++n, p += 16) {
			uint64_t const child = u64(p + 8); // (key n, child n, key n + 1, ...)
			if (level > 0) {
				walk(child, names, depth + 1);
				continue;
			}
			if (!tag(child, "SNOD")) {
				throw std::runtime_error("hdf5: symbol-table node expected");
			}
			uint32_t const nsym = u16(child + 6);
			for (uint32_t s = 0; s < nsym; ++s) {
				uint64_t const e = child + 8 + 40 * static_cast<uint64_t>(s);
				uint64_t const nameAt = names + u64(e);
				need(nameAt, 1);
				std::string const name(reinterpret_cast<char const *>(d_.data() + nameAt), strnlen(reinterpret_cast<char const *>(d_.data() + nameAt), d_.size() - nameAt));
				objects_[name] = u64(e + 8);
			}
		}
	}
	[[nodiscard]] auto messages(uint64_t hdr) const -> std::vector<Msg>
	{
		need(hdr, 16);
		if (d_[hdr] != 1) {
			throw std::runtime_error("hdf5: only version-1 object headers are read");
		}
		uint32_t const nmsg = u16(hdr + 2);
		std::vector<std::pair<uint64_t, uint64_t>> blocks{{hdr + 16, u32(hdr + 8)}};
		std::vector<Msg> out;
		for (size_t blk = 0; blk < blocks.size() && out.size() < nmsg; ++blk) {
			uint64_t p = blocks[blk].first;
			uint64_t const end = p + blocks[blk].second;
			need(p, blocks[blk].second);
			while (p + 8 <= end && out.size() < nmsg) {
				int const type = static_cast<int>(u16(p));
				size_t const size = u16(p + 2);
				need(p + 8, size);
				if (type == 0x10) { // continuation
					blocks.emplace_back(u64(p + 8), u64(p + 16));
				}
				out.push_back({type, static_cast<size_t>(p + 8), size});
				p += 8 + size;
			}
		}
		return out;
	}
	static auto dataspace(unsigned char const *b, size_t size) -> std::vector<uint64_t>
	{
		if (size < 4 || (b[0] != 1 && b[0] != 2)) {
			throw std::runtime_error("hdf5: dataspace message of an unknown version");
		}
		int const rank = b[1];
		size_t const off = (b[0] == 1) ? 8 : 4;
		if (off + 8 * static_cast<size_t>(rank) > size) {
			throw std::runtime_error("hdf5: dataspace message shorter than its rank says");
		}
		std::vector<uint64_t> dims(static_cast<size_t>(rank));
		for (int r = 0; r < rank; ++r) {
			dims[static_cast<size_t>(r)] = rd64(b + off + 8 * static_cast<size_t>(r));
		}
		return dims;
	}
	static auto datatype(unsigned char const *b, size_t size) -> Datatype
	{
		if (size < 8) {
			throw std::runtime_error("hdf5: short datatype message");
		}
		Datatype t;
		t.cls = b[0] & 0x0f;
		t.bigEndian = (b[1] & 1) != 0;
		t.isSigned = (b[1] & 8) != 0;
		t.size = static_cast<int>(rd32(b + 4));
		return t;
	}
};

} // namespace h5
} // namespace qk

#endif // QK_HDF5_MINI_HPP_
