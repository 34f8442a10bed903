// TEST INFRASTRUCTURE ONLY (never part of the product library).
// Declaration-only stand-in for the un-vendored third-party/cereal submodule so that the
// reference's own sources under /root/reference compile UNMODIFIED into oracle/_ref/.
// Serialization is not on the DCRTPoly hot path; every archive call throws.
#ifndef ORACLE_CEREAL_STUB_HPP
#define ORACLE_CEREAL_STUB_HPP
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace cereal {
using size_type = uint64_t;

struct Exception : public std::runtime_error {
    explicit Exception(const std::string& w) : std::runtime_error(w) {}
    explicit Exception(const char* w) : std::runtime_error(w) {}
};

template <class T>
struct NameValuePair {
    const char* name;
    T value;
};
template <class T>
inline NameValuePair<T> make_nvp(const char* n, T&& v) {
    return {n, std::forward<T>(v)};
}
template <class T>
inline NameValuePair<T> make_nvp(const std::string& n, T&& v) {
    return {n.c_str(), std::forward<T>(v)};
}
template <class T>
struct BinaryData {
    T data;
    uint64_t size;
};
template <class T>
inline BinaryData<T> binary_data(T&& d, size_t s) {
    return {std::forward<T>(d), static_cast<uint64_t>(s)};
}
template <class T>
struct SizeTag {
    T size;
};
template <class T>
inline SizeTag<T> make_size_tag(T&& s) {
    return {std::forward<T>(s)};
}
template <class B>
struct base_class {
    template <class D>
    explicit base_class(D const*) {}
};
template <class B>
struct virtual_base_class {
    template <class D>
    explicit virtual_base_class(D const*) {}
};

namespace traits {
template <class A>
struct is_text_archive : std::false_type {};
}  // namespace traits

#define ORACLE_STUB_ARCHIVE(NAME, STREAM)                                        \
    class NAME {                                                                 \
    public:                                                                      \
        explicit NAME(STREAM&) {}                                                \
        template <class... Ts>                                                   \
        NAME& operator()(Ts&&...) {                                              \
            throw Exception("cereal stub: serialization is not available");      \
        }                                                                        \
    };
ORACLE_STUB_ARCHIVE(PortableBinaryOutputArchive, std::ostream)
ORACLE_STUB_ARCHIVE(JSONOutputArchive, std::ostream)
ORACLE_STUB_ARCHIVE(PortableBinaryInputArchive, std::istream)
ORACLE_STUB_ARCHIVE(JSONInputArchive, std::istream)
#undef ORACLE_STUB_ARCHIVE
}  // namespace cereal

#define CEREAL_SAVE_FUNCTION_NAME save
#define CEREAL_LOAD_FUNCTION_NAME load
#define CEREAL_CLASS_VERSION(...)
#define CEREAL_REGISTER_TYPE(...)
#define CEREAL_REGISTER_POLYMORPHIC_RELATION(...)
#define CEREAL_REGISTER_DYNAMIC_INIT(...)
#define CEREAL_FORCE_DYNAMIC_INIT(...)
#endif
