// cilqr_config.hpp — C++ counterpart of the reference's GlobalConfig key/value store
// (/root/reference/include/global_config.hpp:30-36, src/global_config.cpp:17-131) for host programs
// that drive the C-ABI without yaml-cpp.  Reads the flattened scenario files shipped under
// toy-example-of-ilqr_amd/scenarios/*.json ("section/key": value, the key set of global_config.cpp:22-92).
// get_config<T>(key) returns T() and reports on stderr for a missing key, as upstream.
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cilqr_amd {

class FlatConfig {
  public:
    static FlatConfig load(const std::string& path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        FlatConfig c;
        c.parse(ss.str());
        return c;
    }

    bool has_key(const std::string& key) const { return raw_.count(key) != 0; }

    template <class T>
    T get_config(const std::string& key) const {
        auto it = raw_.find(key);
        if (it == raw_.end()) {
            std::cerr << "Key not found: " << key << std::endl;
            return T();
        }
        return convert<T>(it->second);
    }

  private:
    std::map<std::string, std::string> raw_;  // key -> raw JSON value text

    static void skip_ws(const std::string& s, size_t& i) {
        while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i;
    }
    static std::string parse_string(const std::string& s, size_t& i) {
        if (s[i] != '"') throw std::runtime_error("config: expected string");
        size_t j = s.find('"', i + 1);
        std::string out = s.substr(i + 1, j - i - 1);
        i = j + 1;
        return out;
    }
    static std::string parse_value_text(const std::string& s, size_t& i) {
        size_t start = i;
        if (s[i] == '"') {
            parse_string(s, i);
        } else if (s[i] == '[') {
            int depth = 0;
            do {
                if (s[i] == '[') ++depth;
                if (s[i] == ']') --depth;
                ++i;
            } while (depth > 0);
        } else {
            while (i < s.size() && s[i] != ',' && s[i] != '}' && !std::isspace(static_cast<unsigned char>(s[i]))) ++i;
        }
        return s.substr(start, i - start);
    }
    void parse(const std::string& s) {
        size_t i = 0;
        skip_ws(s, i);
        if (s[i] != '{') throw std::runtime_error("config: expected object");
        ++i;
        for (;;) {
            skip_ws(s, i);
            if (s[i] == '}') break;
            std::string key = parse_string(s, i);
            skip_ws(s, i);
            if (s[i] != ':') throw std::runtime_error("config: expected ':'");
            ++i;
            skip_ws(s, i);
            raw_[key] = parse_value_text(s, i);
            skip_ws(s, i);
            if (s[i] == ',') ++i;
        }
    }
    static std::vector<double> numbers(const std::string& v) {
        std::vector<double> out;
        size_t i = 0;
        while (i < v.size()) {
            if (std::isdigit(static_cast<unsigned char>(v[i])) || v[i] == '-' || v[i] == '+' || v[i] == '.') {
                char* end = nullptr;
                out.push_back(std::strtod(v.c_str() + i, &end));
                i = static_cast<size_t>(end - v.c_str());
            } else {
                ++i;
            }
        }
        return out;
    }
    template <class T>
    static T convert(const std::string& v);
};

template <>
inline double FlatConfig::convert<double>(const std::string& v) { return std::strtod(v.c_str(), nullptr); }
template <>
inline int FlatConfig::convert<int>(const std::string& v) { return static_cast<int>(std::strtod(v.c_str(), nullptr)); }
template <>
inline bool FlatConfig::convert<bool>(const std::string& v) { return v == "true"; }
template <>
inline std::string FlatConfig::convert<std::string>(const std::string& v) {
    return (v.size() >= 2 && v.front() == '"') ? v.substr(1, v.size() - 2) : v;
}
template <>
inline std::vector<double> FlatConfig::convert<std::vector<double>>(const std::string& v) { return numbers(v); }
template <>
inline std::vector<std::vector<double>> FlatConfig::convert<std::vector<std::vector<double>>>(const std::string& v) {
    std::vector<std::vector<double>> out;
    size_t i = 1;  // skip the outer '['
    while (i < v.size()) {
        if (v[i] == '[') {
            size_t j = v.find(']', i);
            out.push_back(numbers(v.substr(i, j - i + 1)));
            i = j + 1;
        } else {
            ++i;
        }
    }
    return out;
}

}  // namespace cilqr_amd
