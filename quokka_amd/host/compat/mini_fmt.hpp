// fmt::format for the handful of uses in the reference's problem files ("{}", "{:.4f}", "{:d}", "{:e}" ...): each replacement field is
// turned into the printf conversion it stands for and formatted with snprintf.  libfmt is an un-vendored submodule of the reference.
#ifndef QK_HOST_MINI_FMT_HPP_
#define QK_HOST_MINI_FMT_HPP_
#include <cstdio>
#include <sstream>
#include <string>
#include <type_traits>

namespace fmt
{
namespace detail
{
template <typename T> auto one(std::string const &spec, T const &v) -> std::string
{
	if constexpr (std::is_arithmetic_v<T>) {
		if (!spec.empty()) {
			std::string f = "%" + spec;
			char const last = spec.back();
			if (std::is_floating_point_v<T>) {
				if (last != 'f' && last != 'e' && last != 'g' && last != 'E' && last != 'G') {
					f += 'g';
				}
				char buf[128];
				std::snprintf(buf, sizeof(buf), f.c_str(), static_cast<double>(v));
				return buf;
			}
			if (last == 'd') {
				f.pop_back();
			}
			f += "lld";
			char buf[128];
			std::snprintf(buf, sizeof(buf), f.c_str(), static_cast<long long>(v));
			return buf;
		}
	}
	std::ostringstream s;
	s.precision(17);
	s << v;
	return s.str();
}
inline void fill(std::string &out, std::string const &f, size_t pos) { out += f.substr(pos); }
template <typename T, typename... R> void fill(std::string &out, std::string const &f, size_t pos, T const &v, R const &...rest)
{
	size_t const a = f.find('{', pos);
	if (a == std::string::npos) {
		out += f.substr(pos);
		return;
	}
	size_t const b = f.find('}', a);
	out += f.substr(pos, a - pos);
	std::string spec = f.substr(a + 1, b - a - 1);
	if (!spec.empty() && spec[0] == ':') {
		spec = spec.substr(1);
	}
	out += one(spec, v);
	fill(out, f, b + 1, rest...);
}
} // namespace detail
template <typename... A> auto format(std::string const &f, A const &...args) -> std::string
{
	std::string out;
	detail::fill(out, f, 0, args...);
	return out;
}
} // namespace fmt
#endif
