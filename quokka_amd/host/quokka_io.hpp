// quokka_io.hpp — the on-disk formats of AMRSimulation for the C++ host mirror:
//   plotfiles      AMRSimulation::WritePlotFile -> amrex::WriteMultiLevelPlotfile + metadata.yaml   reference src/simulation.hpp:2294-2336
//   checkpoints    AMRSimulation::WriteCheckpointFile / ReadCheckpointFile                          reference src/simulation.hpp:2564-2834
// The checkpoint `Header` is written by the reference's own code and restated from it.  The plotfile `Header`, the `Cell_H` MultiFab
// header and the FAB header belong to AMReX (not vendored); the reference carries its own writer of the same three pieces for 2-D
// slices (src/io/DiagFramePlane.cpp:321-386 `Write2DPlotfileHeader`, :517-572 `Write2DMFHeader`, :691-699 `write_2D_header`,
// :575-689 `Find2FOffsets`), which is what this file follows, with all AMREX_SPACEDIM directions kept.  Text produced inside AMReX
// by `operator<<` (Box, FabOnDisk, the RealDescriptor of IEEE little-endian doubles, BoxArray::writeOn) is restated from its
// published form: unpinned, readable back by the reader below and by quokka_amd/plotfile.py.
// Host-side I/O, off the timed path: fabs are staged through host buffers.
#ifndef QK_HOST_QUOKKA_IO_HPP_
#define QK_HOST_QUOKKA_IO_HPP_

#include <chrono>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "amrex_mini.hpp"

namespace quokka::io
{

// amrex::Concatenate(root, num, mindigits)
inline auto Concatenate(std::string const &root, int num, int mindigits = 5) -> std::string
{
	std::ostringstream s;
	s << root << std::setw(mindigits) << std::setfill('0') << num;
	return s.str();
}

// operator<<(std::ostream&, amrex::Box const&): ((lo) (hi) (type)), cell-centred type = 0 in each direction; a face-centred box (`b` is then
// the cell box it belongs to) is nodal in `facedir`: one more index there, type 1
inline void printBox(std::ostream &os, amrex::Box const &cellBox, int facedir = -1)
{
	amrex::Box b = cellBox;
	if (facedir >= 0) {
		b.hi[facedir] += 1;
	}
	auto vec = [&](int const *v) {
		os << '(';
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			os << (d > 0 ? "," : "") << v[d];
		}
		os << ')';
	};
	int const type[3] = {facedir == 0 ? 1 : 0, facedir == 1 ? 1 : 0, facedir == 2 ? 1 : 0};
	os << '(';
	vec(b.lo);
	os << ' ';
	vec(b.hi);
	os << ' ';
	vec(type);
	os << ')';
}

inline auto readBox(std::istream &is) -> amrex::Box
{
	amrex::Box b;
	auto expect = [&](char c) {
		char got = 0;
		is >> got;
		if (got != c) {
			amrex::Abort(std::string("quokka::io::readBox: expected '") + c + "', found '" + got + "'");
		}
	};
	auto vec = [&](int *v) {
		expect('(');
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			if (d > 0) {
				expect(',');
			}
			is >> v[d];
		}
		expect(')');
	};
	int type[3] = {0, 0, 0};
	expect('(');
	vec(b.lo);
	vec(b.hi);
	vec(type);
	expect(')');
	return b;
}

// amrex::BoxArray::writeOn / readFrom
inline void writeBoxArray(std::ostream &os, std::vector<amrex::Box> const &ba, int facedir = -1)
{
	os << '(' << ba.size() << ' ' << 0 << '\n';
	for (auto const &b : ba) {
		printBox(os, b, facedir);
		os << '\n';
	}
	os << ')';
}

inline auto readBoxArray(std::istream &is) -> std::vector<amrex::Box>
{
	char c = 0;
	is >> c;
	if (c != '(') {
		amrex::Abort("quokka::io::readBoxArray: expected '('");
	}
	long n = 0, hash = 0;
	is >> n >> hash;
	std::vector<amrex::Box> ba;
	for (long i = 0; i < n; ++i) {
		ba.push_back(readBox(is));
	}
	is >> c;
	if (c != ')') {
		amrex::Abort("quokka::io::readBoxArray: expected ')'");
	}
	return ba;
}

// amrex::FPC::NativeRealDescriptor() of an IEEE-754 little-endian double, as printed by operator<<
inline auto nativeRealDescriptor() -> std::string { return "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"; }

// amrex::UtilRenameDirectoryToOld + PreBuildDirectorHierarchy: an existing directory of that name is kept as <name>.old.<digits>
inline void preBuildDirectoryHierarchy(std::string const &name, std::string const &levelPrefix, int nlevels)
{
	namespace fs = std::filesystem;
	if (fs::exists(name)) {
		auto const ticks = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
		std::string old;
		for (int salt = 0;; ++salt) {
			old = name + ".old." + Concatenate("", static_cast<int>((ticks + salt) % 10000000), 7);
			if (!fs::exists(old)) {
				break;
			}
		}
		fs::rename(name, old);
	}
	fs::create_directories(name);
	for (int l = 0; l < nlevels; ++l) {
		fs::create_directories(name + "/" + levelPrefix + std::to_string(l));
	}
}

// Several ranks: the boxes of the whole level and the rank each one lives on (a MultiFab of this mirror holds the local ones, in the order of
// their global indices).  VisMF::Write with NFiles = number of ranks: rank r writes <prefix>_D_<r> with its fabs, rank 0 the header, whose
// FabOnDisk offsets follow from the box list alone (fab header + ncomp x points doubles) and whose minima / maxima are reduced over the ranks.
struct Distribution {
	std::vector<amrex::Box> all;
	std::vector<int> owner;
	[[nodiscard]] auto several() const -> bool { return qkhost::Comm::get().size > 1; }
};

// `FAB <RealDescriptor><box> <ncomp>\n`
inline auto fabHeader(amrex::Box out, int facedir, int nc) -> std::string
{
	std::ostringstream hss;
	hss << "FAB " << nativeRealDescriptor();
	if (facedir >= 0) {
		out.hi[facedir] -= 1;
	}
	printBox(hss, out, facedir);
	hss << ' ' << nc << '\n';
	return hss.str();
}

// amrex::VisMF::Write(mf, prefix): <prefix>_H (Version_v1, How::NFiles, per-fab minima and maxima) and <prefix>_D_00000 holding every
// fab (one rank writes one file), each as `FAB <RealDescriptor><box> <ncomp>\n` followed by the native doubles, component outermost.
// with_ghost = false strips the ghost cells (amrex::WriteMultiLevelPlotfile copies to a MultiFab without ghost cells first).
inline void VisMFWrite(amrex::MultiFab const &mf, std::string const &prefix, bool with_ghost, Distribution const *dist = nullptr)
{
	int const nc = mf.nComp();
	int const ng = with_ghost ? mf.nGrow() : 0;
	bool const several = dist != nullptr && dist->several();
	int const myRank = several ? qkhost::Comm::get().rank : 0;
	std::string const dataName = prefix + "_D_" + Concatenate("", myRank, 5);
	std::string const baseName = std::filesystem::path(dataName).filename().string();
	std::ofstream data(dataName, std::ofstream::out | std::ofstream::trunc | std::ofstream::binary);
	if (!data.good()) {
		amrex::Abort("quokka::io::VisMFWrite: cannot open " + dataName);
	}
	std::vector<long> offsets;
	std::vector<std::vector<double>> mins, maxs;
	for (int b = 0; b < mf.size(); ++b) {
		auto h = mf.copyToHost(b);
		amrex::Box const src = mf.fabbox(b);
		amrex::Box out = mf.validbox(b);
		if (mf.faceDir() >= 0) { // the stored box of a face-centred fab: nodal in that direction (one more index)
			out.hi[mf.faceDir()] += 1;
		}
		out = amrex::grow(out, ng);
		std::vector<double> buf(static_cast<size_t>(out.numPts()) * nc);
		amrex::Array4<double> s(h.data(), src, nc), d(buf.data(), out, nc);
		std::vector<double> mn(nc, std::numeric_limits<double>::max()), mx(nc, std::numeric_limits<double>::lowest());
		for (int n = 0; n < nc; ++n) {
			amrex::HostFor(out, [&](int i, int j, int k) {
				double const v = s(i, j, k, n);
				d(i, j, k, n) = v;
				mn[n] = std::min(mn[n], v);
				mx[n] = std::max(mx[n], v);
			});
		}
		offsets.push_back(static_cast<long>(data.tellp()));
		data << fabHeader(out, mf.faceDir(), nc);
		data.write(reinterpret_cast<char const *>(buf.data()), static_cast<std::streamsize>(sizeof(double) * buf.size()));
		mins.push_back(mn);
		maxs.push_back(mx);
	}
	data.close();

	std::vector<amrex::Box> const &headerBoxes = several ? dist->all : mf.boxArray();
	std::vector<std::string> fileOf(headerBoxes.size(), baseName);
	if (several) { // the tables of the header for ALL boxes: offsets from the box list, minima / maxima reduced over the ranks
		auto &comm = qkhost::Comm::get();
		size_t const nb = dist->all.size();
		std::vector<long> off(nb, 0), next(static_cast<size_t>(comm.size), 0);
		std::vector<double> mn(nb * nc, std::numeric_limits<double>::max()), mx(nb * nc, std::numeric_limits<double>::lowest());
		size_t local = 0;
		std::string const stem = std::filesystem::path(prefix).filename().string();
		for (size_t n = 0; n < nb; ++n) {
			amrex::Box out = dist->all[n];
			if (mf.faceDir() >= 0) {
				out.hi[mf.faceDir()] += 1;
			}
			out = amrex::grow(out, ng);
			int const r = dist->owner[n];
			off[n] = next[r];
			next[r] += static_cast<long>(fabHeader(out, mf.faceDir(), nc).size()) + static_cast<long>(sizeof(double)) * out.numPts() * nc;
			fileOf[n] = stem + "_D_" + Concatenate("", r, 5);
			if (r == comm.rank) {
				if (local >= offsets.size() || offsets[local] != off[n]) {
					amrex::Abort("quokka::io::VisMFWrite: the local fabs are not the boxes this rank owns, in their global order");
				}
				std::copy(mins[local].begin(), mins[local].end(), mn.begin() + static_cast<long>(n * nc));
				std::copy(maxs[local].begin(), maxs[local].end(), mx.begin() + static_cast<long>(n * nc));
				++local;
			}
		}
		comm.allReduceMany(mn.data(), mn.size(), qkhost::Comm::Op::min);
		comm.allReduceMany(mx.data(), mx.size(), qkhost::Comm::Op::max);
		if (comm.rank != 0) {
			return;
		}
		offsets = off;
		mins.assign(nb, std::vector<double>(nc));
		maxs.assign(nb, std::vector<double>(nc));
		for (size_t n = 0; n < nb; ++n) {
			std::copy(mn.begin() + static_cast<long>(n * nc), mn.begin() + static_cast<long>((n + 1) * nc), mins[n].begin());
			std::copy(mx.begin() + static_cast<long>(n * nc), mx.begin() + static_cast<long>((n + 1) * nc), maxs[n].begin());
		}
	}

	std::ofstream hdr(prefix + "_H", std::ios::out | std::ios::trunc);
	if (!hdr.good()) {
		amrex::Abort("quokka::io::VisMFWrite: cannot open " + prefix + "_H");
	}
	hdr.setf(std::ios::floatfield, std::ios::scientific);
	hdr << 1 << '\n';  // VisMF::Header::Version_v1
	hdr << 1 << '\n';  // VisMF::How::NFiles
	hdr << nc << '\n'; // m_ncomp
	hdr << ng << '\n'; // m_ngrow (same in every direction)
	writeBoxArray(hdr, headerBoxes, mf.faceDir());
	hdr << '\n';
	hdr << headerBoxes.size() << '\n';
	for (size_t b = 0; b < headerBoxes.size(); ++b) {
		hdr << "FabOnDisk: " << fileOf[b] << ' ' << offsets[b] << '\n';
	}
	hdr << '\n';
	hdr.precision(16);
	for (auto const *mm : {&mins, &maxs}) {
		hdr << mm->size() << "," << nc << '\n';
		for (auto const &row : *mm) {
			for (double const v : row) {
				hdr << v << ",";
			}
			hdr << "\n";
		}
		if (mm == &mins) {
			hdr << "\n";
		}
	}
}

struct VisMFData {
	int ncomp = 0, nghost = 0;
	std::vector<amrex::Box> boxes;	       // valid boxes (the BoxArray of the header)
	std::vector<amrex::Box> fabboxes;      // as stored (grown by nghost)
	std::vector<std::vector<double>> fabs; // component outermost
};

// amrex::VisMF::Read(mf, prefix)
// wanted: the boxes the caller will copy into.  An entry of the file whose stored box (valid box grown by m_ngrow) meets none of them is
// not opened: on restart each rank reads the fabs under its own boxes, not the whole level.
inline auto VisMFRead(std::string const &prefix, std::vector<amrex::Box> const *wanted = nullptr) -> VisMFData
{
	VisMFData r;
	std::ifstream hdr(prefix + "_H");
	if (!hdr.good()) {
		amrex::Abort("quokka::io::VisMFRead: cannot open " + prefix + "_H");
	}
	int vers = 0, how = 0;
	hdr >> vers >> how >> r.ncomp;
	// m_ngrow: AMReX writes one integer when the count is the same in every direction and the IntVect "(a,b,c)" otherwise (VisMF::Header's
	// operator<<); both are read, an anisotropic count is refused
	hdr >> std::ws;
	if (hdr.peek() == '(') {
		std::string iv;
		std::getline(hdr, iv);
		int g[3] = {0, 0, 0};
		int const n = std::sscanf(iv.c_str(), "(%d,%d,%d)", &g[0], &g[1], &g[2]);
		for (int d = 1; d < n; ++d) {
			if (g[d] != g[0]) {
				amrex::Abort("quokka::io::VisMFRead: anisotropic ghost-cell counts are not supported: " + iv);
			}
		}
		if (n < 1) {
			amrex::Abort("quokka::io::VisMFRead: malformed m_ngrow " + iv);
		}
		r.nghost = g[0];
	} else {
		hdr >> r.nghost;
	}
	r.boxes = readBoxArray(hdr);
	long nfod = 0;
	hdr >> nfod;
	std::string const dir = std::filesystem::path(prefix).parent_path().string();
	for (long n = 0; n < nfod; ++n) {
		std::string tag, name;
		long head = 0;
		hdr >> tag >> name >> head;
		if (tag != "FabOnDisk:") {
			amrex::Abort("quokka::io::VisMFRead: malformed FabOnDisk entry in " + prefix + "_H");
		}
		if (wanted != nullptr && static_cast<size_t>(n) < r.boxes.size()) {
			amrex::Box const stored = amrex::grow(r.boxes[n], r.nghost);
			bool meets = false;
			for (auto const &w : *wanted) {
				meets = meets || (stored & w).ok();
			}
			if (!meets) {
				r.fabboxes.push_back(stored);
				r.fabs.emplace_back(); // not read
				continue;
			}
		}
		std::ifstream data(dir + "/" + name, std::ifstream::in | std::ifstream::binary);
		if (!data.good()) {
			amrex::Abort("quokka::io::VisMFRead: cannot open " + dir + "/" + name);
		}
		data.seekg(head);
		std::string fab;
		data >> fab;
		if (fab != "FAB") {
			amrex::Abort("quokka::io::VisMFRead: no FAB header at offset " + std::to_string(head) + " of " + name);
		}
		// the RealDescriptor: two parenthesised groups inside one pair of parentheses
		int depth = 0;
		std::string desc;
		char c = 0;
		while (data.get(c)) {
			if (c == '(') {
				++depth;
			}
			if (depth > 0) {
				desc.push_back(c);
			}
			if (c == ')' && --depth == 0) {
				break;
			}
		}
		if (desc != nativeRealDescriptor()) {
			amrex::Abort("quokka::io::VisMFRead: only native little-endian doubles are supported, found " + desc);
		}
		amrex::Box const fb = readBox(data);
		int nc = 0;
		data >> nc;
		data.get(c); // the newline that ends the FAB header
		if (nc != r.ncomp || c != '\n') {
			amrex::Abort("quokka::io::VisMFRead: FAB header does not match the MultiFab header");
		}
		std::vector<double> buf(static_cast<size_t>(fb.numPts()) * nc);
		data.read(reinterpret_cast<char *>(buf.data()), static_cast<std::streamsize>(sizeof(double) * buf.size()));
		if (!data.good()) {
			amrex::Abort("quokka::io::VisMFRead: short read in " + name);
		}
		r.fabboxes.push_back(fb);
		r.fabs.push_back(std::move(buf));
	}
	return r;
}

// VisMF::Read into a temporary + ParallelCopy(tmp, 0, 0, ncomp, nghost, nghost) (reference src/simulation.hpp:2795-2801): the file's
// boxes need not be the destination's.  Cells of the destination (ghost cells included) take the file's valid data where it exists,
// the file's ghost data elsewhere.
inline void VisMFReadInto(amrex::MultiFab &dst, std::string const &prefix)
{
	std::vector<amrex::Box> mine;
	for (int b = 0; b < dst.size(); ++b) {
		mine.push_back(dst.fabbox(b));
	}
	VisMFData const src = VisMFRead(prefix, &mine);
	if (src.ncomp != dst.nComp()) {
		amrex::Abort("quokka::io::VisMFReadInto: component count of " + prefix + " does not match");
	}
	int const nc = src.ncomp;
	for (int b = 0; b < dst.size(); ++b) {
		auto h = dst.copyToHost(b);
		amrex::Array4<double> d(h.data(), dst.fabbox(b), nc);
		for (int pass = 0; pass < 2; ++pass) { // 0: everything stored, 1: valid cells on top
			for (size_t f = 0; f < src.fabs.size(); ++f) {
				if (src.fabs[f].empty()) {
					continue; // outside every box of this rank
				}
				amrex::Box const &from = (pass == 0) ? src.fabboxes[f] : src.boxes[f];
				amrex::Box isect;
				bool ok = true;
				for (int a = 0; a < 3; ++a) {
					isect.lo[a] = std::max(from.lo[a], dst.fabbox(b).lo[a]);
					isect.hi[a] = std::min(from.hi[a], dst.fabbox(b).hi[a]);
					ok = ok && isect.lo[a] <= isect.hi[a];
				}
				if (!ok) {
					continue;
				}
				amrex::Array4<const double> s(src.fabs[f].data(), src.fabboxes[f], nc);
				for (int n = 0; n < nc; ++n) {
					amrex::HostFor(isect, [&](int i, int j, int k) { d(i, j, k, n) = s(i, j, k, n); });
				}
			}
		}
		dst.copyFromHost(b, h);
	}
}

// amrex::WriteMultiLevelPlotfile(name, nlevels, mf, varnames, geom, time, level_steps, ref_ratio): version HyperCLaw-V1.1,
// Level_<l>/Cell.  Header layout as in the reference's Write2DPlotfileHeader (src/io/DiagFramePlane.cpp:321-386).
inline void WriteMultiLevelPlotfile(std::string const &name, int nlevels, std::vector<amrex::MultiFab const *> const &mf, std::vector<std::string> const &varnames,
				    std::vector<amrex::Geometry> const &geom, double time, std::vector<int> const &level_steps, int ref_ratio = 2,
				    std::vector<Distribution> const *dists = nullptr)
{
	std::string const levelPrefix = "Level_", mfPrefix = "Cell";
	// several ranks (dists): rank 0 makes the directories and writes the Header, every rank its own data file per level
	bool const several = dists != nullptr && qkhost::Comm::get().size > 1;
	bool const root = !several || qkhost::Comm::get().rank == 0;
	if (root) {
		preBuildDirectoryHierarchy(name, levelPrefix, nlevels);
	}
	if (several) {
		qkhost::Comm::get().barrier();
	}
	int const finest_level = nlevels - 1;
	std::ofstream Hfile;
	std::ostringstream Hnone;
	if (root) {
		Hfile.open(name + "/Header", std::ofstream::out | std::ofstream::trunc | std::ofstream::binary);
		if (!Hfile.good()) {
			amrex::Abort("quokka::io::WriteMultiLevelPlotfile: cannot open " + name + "/Header");
		}
	}
	std::ostream &H = root ? static_cast<std::ostream &>(Hfile) : static_cast<std::ostream &>(Hnone);
	H.precision(17);
	H << "HyperCLaw-V1.1" << '\n';
	H << varnames.size() << '\n';
	for (auto const &v : varnames) {
		H << v << "\n";
	}
	H << AMREX_SPACEDIM << '\n';
	H << time << '\n';
	H << finest_level << '\n';
	for (int d = 0; d < AMREX_SPACEDIM; ++d) {
		H << geom[0].ProbLo(d) << ' ';
	}
	H << '\n';
	for (int d = 0; d < AMREX_SPACEDIM; ++d) {
		H << geom[0].ProbHiArray()[d] << ' ';
	}
	H << '\n';
	for (int i = 0; i < finest_level; ++i) {
		H << ref_ratio << ' ';
	}
	H << '\n';
	for (int i = 0; i <= finest_level; ++i) {
		printBox(H, geom[i].Domain());
		H << ' ';
	}
	H << '\n';
	for (int i = 0; i <= finest_level; ++i) {
		H << level_steps[i] << ' ';
	}
	H << '\n';
	for (int i = 0; i <= finest_level; ++i) {
		for (int d = 0; d < AMREX_SPACEDIM; ++d) {
			H << geom[i].CellSize(d) << ' ';
		}
		H << '\n';
	}
	H << 0 << '\n'; // Geometry::Coord(): cartesian
	H << "0\n";	// boundary width
	for (int level = 0; level <= finest_level; ++level) {
		auto const &ba = several ? (*dists)[level].all : mf[level]->boxArray();
		H << level << ' ' << ba.size() << ' ' << time << '\n';
		H << level_steps[level] << '\n';
		for (auto const &b : ba) {
			for (int d = 0; d < AMREX_SPACEDIM; ++d) { // RealBox of the box shifted to a domain that starts at index 0
				int const dlo = geom[level].Domain().lo[d];
				H << geom[level].ProbLo(d) + (b.lo[d] - dlo) * geom[level].CellSize(d) << ' '
				  << geom[level].ProbLo(d) + (b.hi[d] - dlo + 1) * geom[level].CellSize(d) << '\n';
			}
		}
		H << levelPrefix << level << '/' << mfPrefix << '\n'; // amrex::MultiFabHeaderPath
	}
	if (root) {
		Hfile.close();
	}
	for (int level = 0; level <= finest_level; ++level) {
		VisMFWrite(*mf[level], name + "/" + levelPrefix + std::to_string(level) + "/" + mfPrefix, /*with_ghost=*/false, several ? &(*dists)[level] : nullptr);
	}
	if (several) {
		qkhost::Comm::get().barrier(); // the file is complete when any rank returns
	}
}

// AMRSimulation::WriteMetadataFile (reference src/simulation.hpp:2338-2366): the YAML map of simulationMetadata_, which is empty
// unless a problem generator fills it (none of the configured problems does): yaml-cpp emits an empty flow map.
inline void WriteMetadataFile(std::string const &path)
{
	std::ofstream f(path, std::ofstream::out | std::ofstream::trunc | std::ofstream::binary);
	f << "{}" << '\n';
}

struct CheckpointHeader {
	int finest_level = 0;
	std::vector<int> istep;
	std::vector<double> dt, tNew;
	std::vector<std::vector<amrex::Box>> grids;
};

// AMRSimulation::WriteCheckpointFile (reference src/simulation.hpp:2564-2666): Header (title, finest_level, istep[], dt[], t_new[],
// one BoxArray per level), metadata.yaml, Level_<l>/Cell = state_new_cc_[l] with its ghost cells, and the `last_chk` symlink
// `faces` (optional): per level the AMREX_SPACEDIM face-centred arrays, written as Level_<l>/Face_x|y|z (reference src/simulation.hpp:2645-2652)
inline void WriteCheckpointFile(std::string const &name, CheckpointHeader const &h, std::vector<amrex::MultiFab const *> const &state,
				std::vector<std::array<amrex::MultiFab const *, AMREX_SPACEDIM>> const &faces = {}, std::vector<Distribution> const *dists = nullptr)
{
	int const nlevels = h.finest_level + 1;
	// several ranks (dists; h.grids then holds the boxes of the whole level): as WriteMultiLevelPlotfile
	bool const several = dists != nullptr && qkhost::Comm::get().size > 1;
	bool const root = !several || qkhost::Comm::get().rank == 0;
	if (root) {
		preBuildDirectoryHierarchy(name, "Level_", nlevels);
	}
	if (several) {
		qkhost::Comm::get().barrier();
	}
	std::ofstream Hfile;
	std::ostringstream Hnone;
	if (root) {
		Hfile.open(name + "/Header", std::ofstream::out | std::ofstream::trunc | std::ofstream::binary);
		if (!Hfile.good()) {
			amrex::Abort("quokka::io::WriteCheckpointFile: cannot open " + name + "/Header");
		}
	}
	std::ostream &H = root ? static_cast<std::ostream &>(Hfile) : static_cast<std::ostream &>(Hnone);
	H.precision(17);
	H << "Checkpoint file for QuokkaCode\n";
	H << h.finest_level << "\n";
	for (int const s : h.istep) {
		H << s << " ";
	}
	H << "\n";
	for (double const v : h.dt) {
		H << v << " ";
	}
	H << "\n";
	for (double const v : h.tNew) {
		H << v << " ";
	}
	H << "\n";
	for (int lev = 0; lev <= h.finest_level; ++lev) {
		writeBoxArray(H, h.grids[lev]);
		H << '\n';
	}
	if (root) {
		Hfile.close();
		WriteMetadataFile(name + "/metadata.yaml");
	}
	for (int lev = 0; lev <= h.finest_level; ++lev) {
		Distribution const *dist = several ? &(*dists)[lev] : nullptr;
		VisMFWrite(*state[lev], name + "/Level_" + std::to_string(lev) + "/Cell", /*with_ghost=*/true, dist);
		if (lev < static_cast<int>(faces.size())) {
			char const *dirName[3] = {"x", "y", "z"};
			for (int d = 0; d < AMREX_SPACEDIM; ++d) {
				VisMFWrite(*faces[lev][d], name + "/Level_" + std::to_string(lev) + "/Face_" + dirName[d], /*with_ghost=*/true, dist);
			}
		}
	}
	if (several) {
		qkhost::Comm::get().barrier();
	}
	if (!root) {
		return;
	}
	// SetLastCheckpointSymlink (reference src/simulation.hpp:2543-2562)
	namespace fs = std::filesystem;
	fs::path const link = fs::path(name).parent_path() / "last_chk";
	std::error_code ec;
	if (fs::is_symlink(link)) {
		fs::remove(link, ec);
	}
	fs::create_directory_symlink(fs::path(name).filename(), link, ec);
}

// the Header part of AMRSimulation::ReadCheckpointFile (reference src/simulation.hpp:2676-2735); istep / dt / t_new hold one entry
// per possible level (max_level + 1), of which the first finest_level + 1 are live
inline auto ReadCheckpointHeader(std::string const &name) -> CheckpointHeader
{
	std::ifstream is(name + "/Header");
	if (!is.good()) {
		amrex::Abort("quokka::io::ReadCheckpointHeader: cannot open " + name + "/Header");
	}
	CheckpointHeader h;
	std::string line, word;
	std::getline(is, line); // title
	is >> h.finest_level;
	std::getline(is, line);
	std::getline(is, line);
	{
		std::istringstream lis(line);
		while (lis >> word) {
			h.istep.push_back(std::stoi(word));
		}
	}
	std::getline(is, line);
	{
		std::istringstream lis(line);
		while (lis >> word) {
			h.dt.push_back(std::stod(word));
		}
	}
	std::getline(is, line);
	{
		std::istringstream lis(line);
		while (lis >> word) {
			h.tNew.push_back(std::stod(word));
		}
	}
	for (int lev = 0; lev <= h.finest_level; ++lev) {
		h.grids.push_back(readBoxArray(is));
		std::getline(is, line);
	}
	return h;
}

} // namespace quokka::io

#endif // QK_HOST_QUOKKA_IO_HPP_
