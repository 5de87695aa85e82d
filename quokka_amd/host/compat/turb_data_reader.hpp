// turb_data_reader.hpp — the reference's src/turbulence/TurbDataReader.{hpp,cpp}: the velocity perturbations a problem seeds its turbulence with,
// three 3-D datasets (/pertx, /perty, /pertz) of an HDF5 file, read here through the library's own reader of the file format
// (csrc/qk_hdf5_mini.hpp; no libhdf5 in this image).  The reference's tests name "zdrv.hdf5", which is generated outside the reference tree: no file
// to test against — the functions are what the problem files need to compile and link, and work for a contiguous float64 file.
#ifndef QK_HOST_COMPAT_TURB_DATA_READER_HPP_
#define QK_HOST_COMPAT_TURB_DATA_READER_HPP_

#include <cmath>
#include <string>

#include "../../csrc/qk_hdf5_mini.hpp"
#include "../amrex_mini.hpp"
#include "mini_fmt.hpp"

using turb_data = struct turb_data {
	amrex::Table3D<double> dvx;
	amrex::Table3D<double> dvy;
	amrex::Table3D<double> dvz;
};

inline auto qk_read_turb_dataset(qk::h5::File const &file, char const *name) -> amrex::Table3D<double>
{
	qk::h5::Dataset const ds = file.dataset(name);
	if (ds.dims.size() != 3) {
		amrex::Abort(std::string("TurbDataReader: /") + name + " is not a 3-D dataset");
	}
	auto const v = qk::h5::asDoubles(ds.type, ds.raw, ds.count());
	auto *data = new double[v.size()]; // (the reference leaks the same array: it lives as long as the table that views it)
	std::copy(v.begin(), v.end(), data);
	amrex::GpuArray<int, 3> const lo{{0, 0, 0}};
	amrex::GpuArray<int, 3> const hi{{static_cast<int>(ds.dims[0]), static_cast<int>(ds.dims[1]), static_cast<int>(ds.dims[2])}};
	return amrex::Table3D<double>(data, lo, hi);
}

inline void initialize_turbdata(turb_data &data, std::string &data_file)
{
	amrex::Print() << "Initializing turbulence data...\n";
	amrex::Print() << "data_file: " << data_file << ".\n";
	try {
		qk::h5::File const file(data_file);
		data.dvx = qk_read_turb_dataset(file, "pertx");
		data.dvy = qk_read_turb_dataset(file, "perty");
		data.dvz = qk_read_turb_dataset(file, "pertz");
	} catch (std::exception const &e) {
		amrex::Abort(std::string("Failed to open data file! ") + e.what());
	}
}

inline auto get_tabledata(amrex::Table3D<double> &in_t) -> amrex::TableData<double, 3>
{
	amrex::Array<int, 3> tlo{in_t.begin[0], in_t.begin[1], in_t.begin[2]};
	amrex::Array<int, 3> thi{in_t.end[0] - 1, in_t.end[1] - 1, in_t.end[2] - 1};
	amrex::TableData<double, 3> tableData(tlo, thi, amrex::The_Pinned_Arena());
	auto h_table = tableData.table();
	for (int i = tlo[0]; i <= thi[0]; ++i) {
		for (int j = tlo[1]; j <= thi[1]; ++j) {
			for (int k = tlo[2]; k <= thi[2]; ++k) {
				h_table(i, j, k) = in_t(i, j, k);
			}
		}
	}
	return tableData;
}

// root mean square of the perturbation's magnitude
inline auto computeRms(amrex::TableData<amrex::Real, 3> &dvx, amrex::TableData<amrex::Real, 3> &dvy, amrex::TableData<amrex::Real, 3> &dvz) -> amrex::Real
{
	auto const tlo = dvx.lo();
	auto const thi = dvx.hi();
	auto const x = dvx.const_table(), y = dvy.const_table(), z = dvz.const_table();
	amrex::Real sum = 0;
	amrex::Long n = 0;
	for (int i = tlo[0]; i <= thi[0]; ++i) {
		for (int j = tlo[1]; j <= thi[1]; ++j) {
			for (int k = tlo[2]; k <= thi[2]; ++k) {
				sum += x(i, j, k) * x(i, j, k) + y(i, j, k) * y(i, j, k) + z(i, j, k) * z(i, j, k);
				++n;
			}
		}
	}
	return std::sqrt(sum / static_cast<amrex::Real>(n));
}

#endif // QK_HOST_COMPAT_TURB_DATA_READER_HPP_
