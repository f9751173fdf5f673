#include "java_format.h"

#include <charconv>
#include <cmath>
#include <cstring>

namespace kcfhost {
namespace {

// shortest round-trip decimal digits: value = 0.d1d2d3... x 10^exp10
template <typename T>
void shortest_digits(T ax, std::string& digits, int& exp10) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, ax, std::chars_format::scientific);
    std::string s(buf, r.ptr);                       // d.ddddde[+-]xx
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    const int ex = std::stoi(s.substr(e + 1));
    digits.clear();
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    exp10 = ex + 1;
}

std::string java_layout(const std::string& digits, int exp10, bool neg) {
    const int e = exp10 - 1;                         // exponent of the d.ddd form
    std::string out = neg ? "-" : "";
    if (e >= -3 && e < 7) {
        if (e >= 0) {
            std::string ip = digits.substr(0, std::min<size_t>(digits.size(), (size_t)e + 1));
            ip.append((size_t)(e + 1) - ip.size(), '0');
            std::string fp = digits.size() > (size_t)e + 1 ? digits.substr((size_t)e + 1) : "0";
            return out + ip + "." + fp;
        }
        return out + "0." + std::string((size_t)(-e - 1), '0') + digits;
    }
    std::string m = digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0");
    return out + m + "E" + std::to_string(e);
}

}  // namespace

std::string java_fmt_2f(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    const bool neg = std::signbit(x);
    const double ax = std::fabs(x);
    if (ax == 0) return neg ? "-0.00" : "0.00";
    std::string d;
    int exp10;
    shortest_digits(ax, d, exp10);
    // decimal string: integer part + fraction digits
    std::string ip, fp;
    if (exp10 <= 0) {
        ip = "0";
        fp = std::string((size_t)(-exp10), '0') + d;
    } else if ((size_t)exp10 >= d.size()) {
        ip = d + std::string((size_t)exp10 - d.size(), '0');
        fp = "";
    } else {
        ip = d.substr(0, (size_t)exp10);
        fp = d.substr((size_t)exp10);
    }
    bool up = fp.size() > 2 && fp[2] >= '5';          // HALF_UP on the third fraction digit
    fp.resize(2, '0');
    std::string num = ip + fp;                        // digits without the point
    if (up) {
        int i = (int)num.size() - 1;
        while (i >= 0) {
            if (num[i] == '9') { num[i] = '0'; --i; } else { num[i]++; break; }
        }
        if (i < 0) num.insert(num.begin(), '1');
    }
    std::string out = num.substr(0, num.size() - 2) + "." + num.substr(num.size() - 2);
    return neg ? "-" + out : out;
}

std::string java_double_to_string(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
    std::string d;
    int exp10;
    shortest_digits(std::fabs(x), d, exp10);
    return java_layout(d, exp10, x < 0);
}

std::string java_float_to_string(float x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
    std::string d;
    int exp10;
    shortest_digits(std::fabs(x), d, exp10);
    return java_layout(d, exp10, x < 0);
}

}  // namespace kcfhost
