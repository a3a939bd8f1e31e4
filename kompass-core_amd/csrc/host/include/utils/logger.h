// Level-filtered logger of the kompass_cpp surface (reference: utils/logger.h:
// LogLevel, setLogLevel, setLogFile and the LOG_* macros).
#pragma once

#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>

namespace Kompass {

enum class LogLevel { DEBUG = 0, INFO = 1, WARNING = 2, ERROR = 3 };

class Logger {
 public:
  static Logger &getInstance() {
    static Logger l;
    return l;
  }
  void setLogLevel(LogLevel l) { level_ = l; }
  void setLogFile(const std::string &path) {
    std::lock_guard<std::mutex> g(mu_);
    file_.close();
    if (!path.empty()) file_.open(path, std::ios::app);
  }
  template <typename... A>
  void log(LogLevel l, const char *tag, A &&...args) {
    if (static_cast<int>(l) < static_cast<int>(level_)) return;
    std::ostringstream os;
    os << "[" << tag << "] ";
    (os << ... << args);
    std::lock_guard<std::mutex> g(mu_);
    (l == LogLevel::ERROR ? std::cerr : std::cout) << os.str() << std::endl;
    if (file_.is_open()) file_ << os.str() << std::endl;
  }

 private:
  LogLevel level_ = LogLevel::WARNING;
  std::ofstream file_;
  std::mutex mu_;
};

inline void setLogLevel(LogLevel l) { Logger::getInstance().setLogLevel(l); }
inline void setLogFile(const std::string &p) { Logger::getInstance().setLogFile(p); }

}  // namespace Kompass

#define LOG_DEBUG(...) ::Kompass::Logger::getInstance().log(::Kompass::LogLevel::DEBUG, "DEBUG", __VA_ARGS__)
#define LOG_INFO(...) ::Kompass::Logger::getInstance().log(::Kompass::LogLevel::INFO, "INFO", __VA_ARGS__)
#define LOG_WARNING(...) ::Kompass::Logger::getInstance().log(::Kompass::LogLevel::WARNING, "WARNING", __VA_ARGS__)
#define LOG_ERROR(...) ::Kompass::Logger::getInstance().log(::Kompass::LogLevel::ERROR, "ERROR", __VA_ARGS__)
