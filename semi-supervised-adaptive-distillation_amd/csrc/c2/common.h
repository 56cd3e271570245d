// common.h -- error handling and small utilities of the Caffe2-shaped surface.
//
// Mirrors the contract of caffe2/core/logging.h in the reference: failed
// CAFFE_ENFORCE* / CAFFE_THROW raise caffe2::EnforceNotMet, whose message the
// operator runner augments with the operator definition
// (caffe2/core/operator.h:369-395).
#ifndef C2HIP_COMMON_H_
#define C2HIP_COMMON_H_

#include <cstdint>
#include <exception>
#include <sstream>
#include <string>
#include <vector>

#define C2HIP_API __attribute__((visibility("default")))

namespace caffe2 {

using TIndex = int64_t;
using std::string;
using std::vector;

inline void MakeStringInternal(std::ostringstream&) {}
template <class T, class... Rest>
inline void MakeStringInternal(std::ostringstream& ss, const T& t, const Rest&... rest) {
  ss << t;
  MakeStringInternal(ss, rest...);
}
template <class... Args>
inline string MakeString(const Args&... args) {
  std::ostringstream ss;
  MakeStringInternal(ss, args...);
  return ss.str();
}

class C2HIP_API EnforceNotMet : public std::exception {
 public:
  EnforceNotMet(const char* file, int line, const char* cond, const string& msg) {
    full_ = MakeString("[enforce fail at ", file, ":", line, "] ", cond, ". ", msg);
  }
  void AppendMessage(const string& m) { full_ += "\n" + m; }
  const char* what() const noexcept override { return full_.c_str(); }
  const string& msg() const { return full_; }

 private:
  string full_;
};

// Thrown by an operator constructor that cannot serve this definition; the
// factory then tries the next engine (caffe2/core/operator.h:765-782).
class C2HIP_API UnsupportedOperatorFeature : public std::exception {
 public:
  explicit UnsupportedOperatorFeature(const string& m) : msg_(m) {}
  const char* what() const noexcept override { return msg_.c_str(); }

 private:
  string msg_;
};

}  // namespace caffe2

#define CAFFE_THROW(...) \
  throw ::caffe2::EnforceNotMet(__FILE__, __LINE__, "", ::caffe2::MakeString(__VA_ARGS__))

#define CAFFE_ENFORCE(cond, ...)                                                    \
  do {                                                                              \
    if (!(cond))                                                                    \
      throw ::caffe2::EnforceNotMet(__FILE__, __LINE__, #cond,                      \
                                    ::caffe2::MakeString(__VA_ARGS__));             \
  } while (0)

#define CAFFE_ENFORCE_BINARY_(a, b, op, ...)                                        \
  do {                                                                              \
    const auto& c2_a_ = (a);                                                        \
    const auto& c2_b_ = (b);                                                        \
    if (!(c2_a_ op c2_b_))                                                          \
      throw ::caffe2::EnforceNotMet(                                                \
          __FILE__, __LINE__, #a " " #op " " #b,                                    \
          ::caffe2::MakeString(c2_a_, " vs ", c2_b_, ". ", ##__VA_ARGS__));         \
  } while (0)

#define CAFFE_ENFORCE_EQ(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, ==, ##__VA_ARGS__)
#define CAFFE_ENFORCE_NE(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, !=, ##__VA_ARGS__)
#define CAFFE_ENFORCE_GE(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, >=, ##__VA_ARGS__)
#define CAFFE_ENFORCE_GT(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, >, ##__VA_ARGS__)
#define CAFFE_ENFORCE_LE(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, <=, ##__VA_ARGS__)
#define CAFFE_ENFORCE_LT(a, b, ...) CAFFE_ENFORCE_BINARY_(a, b, <, ##__VA_ARGS__)

#define CAFFE_NOT_IMPLEMENTED CAFFE_THROW("Not Implemented.")

#endif  // C2HIP_COMMON_H_
