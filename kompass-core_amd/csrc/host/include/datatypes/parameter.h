// Typed, range-checked configuration parameters (reference surface:
// datatypes/parameter.h:15-289 -- Parameter / Parameters with the same names,
// value types int/double/string/bool and std::out_of_range on a violation).
#pragma once

#include <map>
#include <stdexcept>
#include <string>
#include <variant>

class Parameter {
 public:
  using ValueType = std::variant<int, double, std::string, bool>;

  Parameter() : value_(0), lo_(0), hi_(0), bounded_(false) {}
  template <typename T>
  Parameter(T def, T lo, T hi, std::string description = "Parameter")
      : value_(def), lo_(lo), hi_(hi), bounded_(true),
        description_(std::move(description)) {}
  template <typename T>
  Parameter(T def, std::string description = "Parameter")
      : value_(def), lo_(0), hi_(0), bounded_(false),
        description_(std::move(description)) {}

  template <typename T>
  void setValue(T v) {
    // ints may feed double parameters and vice versa (Python hands both)
    if (std::holds_alternative<double>(value_)) {
      if constexpr (std::is_arithmetic_v<T> && !std::is_same_v<T, bool>) {
        check(static_cast<double>(v));
        value_ = static_cast<double>(v);
        return;
      }
    } else if (std::holds_alternative<int>(value_)) {
      if constexpr (std::is_arithmetic_v<T> && !std::is_same_v<T, bool>) {
        check(static_cast<double>(v));
        value_ = static_cast<int>(v);
        return;
      }
    } else if (std::holds_alternative<bool>(value_)) {
      if constexpr (std::is_same_v<T, bool>) {
        value_ = v;
        return;
      }
    } else if (std::holds_alternative<std::string>(value_)) {
      if constexpr (std::is_convertible_v<T, std::string>) {
        value_ = std::string(v);
        return;
      }
    }
    throw std::invalid_argument("Parameter type mismatch");
  }

  template <typename T>
  T getValue() const {
    if (auto p = std::get_if<T>(&value_)) return *p;
    if constexpr (std::is_same_v<T, double>) {
      if (auto p = std::get_if<int>(&value_)) return static_cast<double>(*p);
    }
    if constexpr (std::is_same_v<T, int>) {
      if (auto p = std::get_if<double>(&value_)) return static_cast<int>(*p);
    }
    throw std::invalid_argument("Parameter type mismatch");
  }
  const std::string &getDescription() const { return description_; }

 private:
  void check(double v) const {
    if (!bounded_) return;
    const double lo = asDouble(lo_), hi = asDouble(hi_);
    if (v < lo || v > hi)
      throw std::out_of_range("Value out of range [" + std::to_string(lo) +
                              ", " + std::to_string(hi) + "]");
  }
  static double asDouble(const ValueType &v) {
    if (auto p = std::get_if<int>(&v)) return *p;
    if (auto p = std::get_if<double>(&v)) return *p;
    return 0.0;
  }
  ValueType value_, lo_, hi_;
  bool bounded_;
  std::string description_;
};

class Parameters {
 public:
  virtual ~Parameters() = default;
  std::map<std::string, Parameter> parameters;

  void addParameter(const std::string &name, const Parameter &p) {
    parameters[name] = p;
  }
  template <typename T>
  void setParameter(const std::string &name, T value) {
    auto it = parameters.find(name);
    if (it == parameters.end())
      throw std::invalid_argument("Parameter not found: " + name);
    it->second.setValue(value);
  }
  template <typename T>
  T getParameter(const std::string &name) const {
    auto it = parameters.find(name);
    if (it == parameters.end())
      throw std::invalid_argument("Parameter not found: " + name);
    return it->second.getValue<T>();
  }
};
