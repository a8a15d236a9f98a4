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
    // .json: the flattened files under scenarios/;  .yaml / .yml: the reference's own layout
    // (/root/reference/config/scenario_*.yaml — nested maps by indentation, scalars, inline lists,
    // "- [..]" list items, '#' comments), flattened to the same "section/key" names with the
    // defaults of global_config.cpp applied.
    static FlatConfig load(const std::string& path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        FlatConfig c;
        const bool yaml = path.size() > 4 && (path.rfind(".yaml") == path.size() - 5 || path.rfind(".yml") == path.size() - 4);
        if (yaml) c.parse_yaml(ss.str());
        else c.parse(ss.str());
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
    static std::string trim(const std::string& t) {
        size_t a = 0, b = t.size();
        while (a < b && std::isspace(static_cast<unsigned char>(t[a]))) ++a;
        while (b > a && std::isspace(static_cast<unsigned char>(t[b - 1]))) --b;
        return t.substr(a, b - a);
    }
    static std::string strip_comment(const std::string& line) {
        bool in_str = false;
        for (size_t i = 0; i < line.size(); ++i) {
            if (line[i] == '"') in_str = !in_str;
            if (line[i] == '#' && !in_str) return line.substr(0, i);
        }
        return line;
    }
    void parse_yaml(const std::string& text) {
        std::vector<std::pair<int, std::string>> stack;  // (indent, key) of the open maps
        std::string list_key;                             // key whose value is a block list of "- [..]" items
        std::string list_val;
        auto flush_list = [&]() {
            if (!list_key.empty()) raw_[list_key] = "[" + list_val + "]";
            list_key.clear();
            list_val.clear();
        };
        std::istringstream in(text);
        std::string line;
        while (std::getline(in, line)) {
            line = strip_comment(line);
            if (trim(line).empty()) continue;
            int indent = 0;
            while (indent < static_cast<int>(line.size()) && line[indent] == ' ') ++indent;
            std::string body = trim(line);
            if (body[0] == '-') {  // list item of the innermost open key
                if (!list_key.empty()) list_val += (list_val.empty() ? "" : ", ") + trim(body.substr(1));
                continue;
            }
            flush_list();
            const size_t colon = body.find(':');
            if (colon == std::string::npos) continue;
            const std::string key = trim(body.substr(0, colon));
            const std::string val = trim(body.substr(colon + 1));
            while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
            std::string full;
            for (const auto& e : stack) full += e.second + "/";
            full += key;
            if (val.empty()) {
                stack.emplace_back(indent, key);
                list_key = full;  // becomes a list if "- " items follow, a map otherwise
            } else {
                raw_[full] = val;
            }
        }
        flush_list();
        // maps that turned out not to be lists leave an empty "[]" entry behind: drop those
        for (auto it = raw_.begin(); it != raw_.end();) {
            if (it->second == "[]") it = raw_.erase(it);
            else ++it;
        }
        // defaults of src/global_config.cpp (.as<T>(default) call sites)
        const std::pair<const char*, const char*> defaults[] = {
            {"lqr/alm_rho_init", "1.0"}, {"lqr/alm_gamma", "0.0"}, {"lqr/max_rho", "100.0"}, {"lqr/max_mu", "1000.0"},
            {"vehicle/reference_point", "\"gravity_center\""}, {"visualization/show_reference_line", "false"},
            {"visualization/show_obstacle_boundary", "false"}};
        for (const auto& d : defaults)
            if (!raw_.count(d.first)) raw_[d.first] = d.second;
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
