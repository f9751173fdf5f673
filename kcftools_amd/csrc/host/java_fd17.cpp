// java_fd17.cpp -- the decimal digits JDK 17 derives from a double / float (jdk.internal.math.FloatingDecimal, the algorithm
// behind Double.toString, Float.toString and -- through FormattedFloatingDecimal -- String.format("%.2f") up to JDK 18).
//
// Why it exists: kcftools targets Java 17 (pom.xml:16), whose digit generator is NOT always the shortest round-tripping one
// (JDK-4511638, fixed in JDK 19), while java_format.cpp prints from the shortest digits. A restatement of the JDK 17 routine
// lets the tests bound the difference: tests/test_host_side.py enumerates the values a KCF file can hold near a two-decimal
// tie (and a few million others) and requires both digit generators to give the same text. Nothing in the product calls this.
//
// Restated from the published algorithm (three arithmetic regimes -- int, long, big integer -- with the same stopping test;
// the int / long regimes wrap around exactly as Java's do, including the "m might overflow" escape), not copied: no JDK
// source is in this repository or in /root/reference.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "java_format.h"

namespace kcfhost {
namespace {

constexpr int EXP_SHIFT = 52;
constexpr uint64_t FRACT_HOB = 1ull << EXP_SHIFT;
constexpr uint64_t SIGNIF_MASK = FRACT_HOB - 1;
constexpr int MAX_SMALL_BIN_EXP = 62, MIN_SMALL_BIN_EXP = -(63 / 3);
const int N_5_BITS[] = {0, 3, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28, 31, 33, 35, 38, 40, 42, 45, 47, 49, 52, 54, 56, 59, 61};
constexpr int N_5 = sizeof N_5_BITS / sizeof N_5_BITS[0];            // also the number of powers of 5 that fit a long
const int INSIGNIFICANT[] = {0,  0,  0,  0,  1,  1,  1,  2,  2,  2,  3,  3,  3,  3,  4,  4,  4,  5,  5,  5,  6,  6,
                             6,  6,  7,  7,  7,  8,  8,  8,  9,  9,  9,  9,  10, 10, 10, 11, 11, 11, 12, 12, 12, 12,
                             13, 13, 13, 14, 14, 14, 15, 15, 15, 15, 16, 16, 16, 17, 17, 17, 18, 18, 18, 19};

int64_t pow5_long(int n) {
    int64_t v = 1;
    for (int i = 0; i < n; ++i) v *= 5;
    return v;
}

// ---- a small unsigned big integer: what the third regime needs (x 5^a 2^b, x 10, compare, add, quotient < 10) ----
struct Big {
    std::vector<uint32_t> w;                                          // little-endian, no leading zero words
    static Big of(uint64_t v) {
        Big b;
        while (v) {
            b.w.push_back((uint32_t)v);
            v >>= 32;
        }
        return b;
    }
    void trim() {
        while (!w.empty() && w.back() == 0) w.pop_back();
    }
    void mul_small(uint32_t m) {
        uint64_t carry = 0;
        for (auto& x : w) {
            const uint64_t t = (uint64_t)x * m + carry;
            x = (uint32_t)t;
            carry = t >> 32;
        }
        if (carry) w.push_back((uint32_t)carry);
    }
    void mul_pow5(int n) {
        for (; n >= 13; n -= 13) mul_small(1220703125u);              // 5^13
        uint32_t m = 1;
        for (int i = 0; i < n; ++i) m *= 5;
        if (m > 1) mul_small(m);
    }
    void shl(int bits) {
        if (w.empty() || bits <= 0) return;
        const int words = bits / 32, r = bits % 32;
        if (r) {
            uint32_t carry = 0;
            for (auto& x : w) {
                const uint32_t n = (x << r) | carry;
                carry = x >> (32 - r);
                x = n;
            }
            if (carry) w.push_back(carry);
        }
        if (words) w.insert(w.begin(), (size_t)words, 0u);
    }
    static int cmp(const Big& a, const Big& b) {
        if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
        for (size_t i = a.w.size(); i-- > 0;)
            if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
        return 0;
    }
    void add(const Big& o) {
        if (w.size() < o.w.size()) w.resize(o.w.size(), 0u);
        uint64_t carry = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            const uint64_t t = (uint64_t)w[i] + (i < o.w.size() ? o.w[i] : 0u) + carry;
            w[i] = (uint32_t)t;
            carry = t >> 32;
        }
        if (carry) w.push_back((uint32_t)carry);
    }
    void sub(const Big& o) {                                          // *this >= o
        int64_t borrow = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            int64_t t = (int64_t)w[i] - (i < o.w.size() ? o.w[i] : 0u) - borrow;
            borrow = t < 0;
            if (t < 0) t += (1ll << 32);
            w[i] = (uint32_t)t;
        }
        trim();
    }
    // q = *this / s (known to be < 10); *this = 10 * (*this mod s)
    int quo_rem_times10(const Big& s) {
        int q = 0;
        while (cmp(*this, s) >= 0) {
            sub(s);
            ++q;
        }
        mul_small(10);
        return q;
    }
    bool zero() const { return w.empty(); }
};

struct Digits {
    std::string d;          // the digits, no leading zeros (one "0" for zero)
    int dec_exponent = 0;   // value = 0.d1d2... x 10^dec_exponent
};

void roundup(Digits& g) {
    int i = (int)g.d.size() - 1;
    char q = g.d[(size_t)i];
    if (q == '9') {
        while (q == '9' && i > 0) {
            g.d[(size_t)i] = '0';
            q = g.d[(size_t)--i];
        }
        if (q == '9') {                                               // carry out: high-order 1, rest 0s, larger exponent
            g.dec_exponent += 1;
            g.d[0] = '1';
            return;
        }
    }
    g.d[(size_t)i] = (char)(q + 1);
}

int estimate_dec_exp(uint64_t fract_bits, int bin_exp) {
    uint64_t b = (0x3FFull << EXP_SHIFT) | (fract_bits & SIGNIF_MASK);
    double d2;
    memcpy(&d2, &b, 8);
    const volatile double d = (d2 - 1.5) * 0.289529654 + 0.176091259 + (double)bin_exp * 0.301029995663981;
    return (int)std::floor(d);
}

void develop_long_digits(Digits& g, int dec_exponent, int64_t lvalue, int insignificant) {
    if (insignificant != 0) {
        const int64_t pow10 = pow5_long(insignificant) << insignificant;
        const int64_t residue = lvalue % pow10;
        lvalue /= pow10;
        dec_exponent += insignificant;
        if (residue >= (pow10 >> 1)) lvalue++;
    }
    std::string rev;
    int c = (int)(lvalue % 10);
    lvalue /= 10;
    while (c == 0) {
        dec_exponent++;
        c = (int)(lvalue % 10);
        lvalue /= 10;
    }
    while (lvalue != 0) {
        rev.push_back((char)('0' + c));
        dec_exponent++;
        c = (int)(lvalue % 10);
        lvalue /= 10;
    }
    rev.push_back((char)('0' + c));
    g.d.assign(rev.rbegin(), rev.rend());
    g.dec_exponent = dec_exponent + 1;
}

// the work horse: bin_exp unbiased, fract_bits normalised (bit 52 set), compat = the stopping rule of Double.toString
Digits dtoa(int bin_exp, uint64_t fract_bits, int n_significant_bits, bool compat) {
    Digits g;
    const int tail_zeros = __builtin_ctzll(fract_bits);
    const int n_fract_bits = EXP_SHIFT + 1 - tail_zeros;
    const int n_tiny_bits = std::max(0, n_fract_bits - bin_exp - 1);
    if (bin_exp <= MAX_SMALL_BIN_EXP && bin_exp >= MIN_SMALL_BIN_EXP) {
        if (n_tiny_bits < N_5 && (n_fract_bits + N_5_BITS[n_tiny_bits]) < 64) {
            if (n_tiny_bits == 0) {                                   // an integer that fits a long
                int insignificant = 0;
                if (bin_exp > n_significant_bits) {
                    const int p2 = bin_exp - n_significant_bits - 1;
                    insignificant = (p2 > 1 && p2 < (int)(sizeof INSIGNIFICANT / sizeof INSIGNIFICANT[0])) ? INSIGNIFICANT[p2] : 0;
                }
                uint64_t v = fract_bits;
                if (bin_exp >= EXP_SHIFT) v <<= (bin_exp - EXP_SHIFT);
                else v >>= (EXP_SHIFT - bin_exp);
                develop_long_digits(g, 0, (int64_t)v, insignificant);
                return g;
            }
        }
    }
    int dec_exp = estimate_dec_exp(fract_bits, bin_exp);
    const int B5 = std::max(0, -dec_exp);
    int B2 = B5 + n_tiny_bits + bin_exp;
    const int S5 = std::max(0, dec_exp);
    int S2 = S5 + n_tiny_bits;
    const int M5 = B5;
    int M2 = B2 - n_significant_bits;
    fract_bits >>= tail_zeros;
    B2 -= n_fract_bits - 1;
    const int common2 = std::min(B2, S2);
    B2 -= common2;
    S2 -= common2;
    M2 -= common2;
    if (n_fract_bits == 1) M2 -= 1;                                   // an exact power of two: the next smaller number is half as far away
    if (M2 < 0) {
        B2 -= M2;
        S2 -= M2;
        M2 = 0;
    }
    bool low, high;
    int64_t low_digit_difference;
    const int Bbits = n_fract_bits + B2 + ((B5 < N_5) ? N_5_BITS[B5] : (B5 * 3));
    const int tenSbits = S2 + 1 + (((S5 + 1) < N_5) ? N_5_BITS[S5 + 1] : ((S5 + 1) * 3));
    if (Bbits < 64 && tenSbits < 64) {
        if (Bbits < 32 && tenSbits < 32) {
            // all ints (Java int arithmetic: wraps around)
            auto mul32 = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); };
            auto shl32 = [](int32_t a, int s) { return (int32_t)((uint32_t)a << (s & 31)); };
            auto add32 = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); };
            int32_t b = shl32(mul32((int32_t)fract_bits, (int32_t)pow5_long(B5)), B2);
            const int32_t s = shl32((int32_t)pow5_long(S5), S2);
            int32_t m = shl32((int32_t)pow5_long(M5), M2);
            const int32_t tens = mul32(s, 10);
            int q = b / s;
            b = mul32(10, b % s);
            m = mul32(m, 10);
            low = b < m;
            high = add32(b, m) > tens;
            if (q == 0 && !high) dec_exp -= 1;
            else g.d.push_back((char)('0' + q));
            if (!compat || dec_exp < -3 || dec_exp >= 8) high = low = false;
            while (!low && !high) {
                q = b / s;
                b = mul32(10, b % s);
                m = mul32(m, 10);
                if (m > 0) {
                    low = b < m;
                    high = add32(b, m) > tens;
                } else {                                              // m has overflowed: certainly > b, and b + m > tens
                    low = true;
                    high = true;
                }
                g.d.push_back((char)('0' + q));
            }
            low_digit_difference = (int64_t)add32(shl32(b, 1), -tens);
        } else {
            auto mul64 = [](int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); };
            auto shl64 = [](int64_t a, int s) { return (int64_t)((uint64_t)a << (s & 63)); };
            auto add64 = [](int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); };
            int64_t b = shl64(mul64((int64_t)fract_bits, pow5_long(B5)), B2);
            const int64_t s = shl64(pow5_long(S5), S2);
            int64_t m = shl64(pow5_long(M5), M2);
            const int64_t tens = mul64(s, 10);
            int q = (int)(b / s);
            b = mul64(10, b % s);
            m = mul64(m, 10);
            low = b < m;
            high = add64(b, m) > tens;
            if (q == 0 && !high) dec_exp -= 1;
            else g.d.push_back((char)('0' + q));
            if (!compat || dec_exp < -3 || dec_exp >= 8) high = low = false;
            while (!low && !high) {
                q = (int)(b / s);
                b = mul64(10, b % s);
                m = mul64(m, 10);
                if (m > 0) {
                    low = b < m;
                    high = add64(b, m) > tens;
                } else {
                    low = true;
                    high = true;
                }
                g.d.push_back((char)('0' + q));
            }
            low_digit_difference = add64(shl64(b, 1), -tens);
        }
    } else {
        Big S = Big::of(1);
        S.mul_pow5(S5);
        S.shl(S2);
        Big B = Big::of(fract_bits);
        B.mul_pow5(B5);
        B.shl(B2);
        Big M = Big::of(1);                                           // 10 x M: the first iteration's multiply is folded in
        M.mul_pow5(M5 + 1);
        M.shl(M2 + 1);
        Big tenS = Big::of(1);
        tenS.mul_pow5(S5 + 1);
        tenS.shl(S2 + 1);
        auto sum_ge = [](const Big& b, const Big& m, const Big& t) {  // b + m >= t
            Big x = b;
            x.add(m);
            return Big::cmp(x, t) >= 0;
        };
        int q = B.quo_rem_times10(S);
        low = Big::cmp(B, M) < 0;
        high = sum_ge(B, M, tenS);
        if (q == 0 && !high) dec_exp -= 1;
        else g.d.push_back((char)('0' + q));
        if (!compat || dec_exp < -3 || dec_exp >= 8) high = low = false;
        while (!low && !high) {
            q = B.quo_rem_times10(S);
            M.mul_small(10);
            low = Big::cmp(B, M) < 0;
            high = sum_ge(B, M, tenS);
            g.d.push_back((char)('0' + q));
        }
        if (high && low) {
            B.shl(1);
            low_digit_difference = Big::cmp(B, tenS);
        } else {
            low_digit_difference = 0;
        }
    }
    g.dec_exponent = dec_exp + 1;
    if (high) {                                                       // the last digit is rounded by the stopping condition
        if (low) {
            if (low_digit_difference == 0) {
                if ((g.d.back() & 1) != 0) roundup(g);                // a tie: to even
            } else if (low_digit_difference > 0) {
                roundup(g);
            }
        } else {
            roundup(g);
        }
    }
    return g;
}

Digits digits_of_double(double ax, bool compat) {                     // ax > 0, finite
    uint64_t bits;
    memcpy(&bits, &ax, 8);
    uint64_t fract = bits & SIGNIF_MASK;
    int bin_exp = (int)((bits >> EXP_SHIFT) & 0x7FF);
    int n_sig;
    if (bin_exp == 0) {                                               // subnormal
        const int lz = __builtin_clzll(fract);
        const int shift = lz - (63 - EXP_SHIFT);
        fract <<= shift;
        bin_exp = 1 - shift;
        n_sig = 64 - lz;
    } else {
        fract |= FRACT_HOB;
        n_sig = EXP_SHIFT + 1;
    }
    return dtoa(bin_exp - 1023, fract, n_sig, compat);
}

Digits digits_of_float(float ax) {                                    // ax > 0, finite; Float.toString is always "compatible"
    uint32_t bits;
    memcpy(&bits, &ax, 4);
    uint32_t fract = bits & 0x7FFFFFu;
    int bin_exp = (int)((bits >> 23) & 0xFF);
    int n_sig;
    if (bin_exp == 0) {
        const int lz = __builtin_clz(fract);
        const int shift = lz - (31 - 23);
        fract <<= shift;
        bin_exp = 1 - shift;
        n_sig = 32 - lz;
    } else {
        fract |= 1u << 23;
        n_sig = 24;
    }
    return dtoa(bin_exp - 127, ((uint64_t)fract) << (EXP_SHIFT - 23), n_sig, true);
}

// FloatingDecimal.BinaryToASCIIBuffer.getChars: the layout of Double.toString / Float.toString
std::string java_chars(const Digits& g, bool neg) {
    std::string out = neg ? "-" : "";
    const int n = (int)g.d.size(), e = g.dec_exponent;
    if (e > 0 && e < 8) {
        const int len = std::min(n, e);
        out += g.d.substr(0, (size_t)len);
        if (len < e) {
            out.append((size_t)(e - len), '0');
            out += ".0";
        } else {
            out += ".";
            out += len < n ? g.d.substr((size_t)len) : "0";
        }
    } else if (e <= 0 && e > -3) {
        out += "0.";
        out.append((size_t)(-e), '0');
        out += g.d;
    } else {
        out += g.d.substr(0, 1);
        out += ".";
        out += n > 1 ? g.d.substr(1) : "0";
        out += "E" + std::to_string(e - 1);
    }
    return out;
}

}  // namespace

std::string jdk17_double_to_string(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
    return java_chars(digits_of_double(std::fabs(x), true), x < 0);
}

std::string jdk17_float_to_string(float x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0.0" : "0.0";
    return java_chars(digits_of_float(std::fabs(x)), x < 0);
}

// String.format("%.2f", x) as JDK 17 does it: FormattedFloatingDecimal.valueOf(x, 2, DECIMAL_FLOAT) -- the digits above with the
// NON-compatible stopping rule, applyPrecision (half-up on the digit behind the precision), fillDecimal
std::string jdk17_fmt_2f(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    const bool neg = std::signbit(x);
    const int precision = 2;
    Digits g;
    if (x == 0) {
        g.d = "0";
        g.dec_exponent = 1;
    } else {
        g = digits_of_double(std::fabs(x), false);
    }
    std::string d = g.d;
    const int n = (int)d.size();
    int exp = g.dec_exponent;
    const int prec = g.dec_exponent + precision;                      // applyPrecision(decExp, digits, nDigits, decExp + precision)
    if (!(prec >= n || prec < 0)) {
        if (prec == 0) {
            if (d[0] >= '5') {
                d.assign((size_t)n, '0');
                d[0] = '1';
                exp += 1;
            } else {
                d.assign((size_t)n, '0');
            }
        } else if (d[(size_t)prec] >= '5') {
            int i = prec;
            char q = d[(size_t)--i];
            bool carried_out = false;
            if (q == '9') {
                while (q == '9' && i > 0) q = d[(size_t)--i];
                if (q == '9') {
                    d.assign((size_t)n, '0');
                    d[0] = '1';
                    exp += 1;
                    carried_out = true;
                }
            }
            if (!carried_out) {
                d[(size_t)i] = (char)(q + 1);
                for (int j = i + 1; j < n; ++j) d[(size_t)j] = '0';
            }
        } else {
            for (int j = prec; j < n; ++j) d[(size_t)j] = '0';
        }
    }
    // fillDecimal(precision, digits, nDigits, exp, isNegative), then Formatter.addZeros up to the precision
    std::string out = neg ? "-" : "";
    std::string frac;
    if (exp > 0) {
        if (n > exp) {                                                // digits on both sides of the point
            out += d.substr(0, (size_t)exp);
            const int t = std::min(n - exp, precision);
            frac = d.substr((size_t)exp, (size_t)t);
        } else {
            out += d;
            out.append((size_t)(exp - n), '0');
        }
    } else {
        const int zeros = std::max(0, std::min(-exp, precision));
        const int t = std::max(0, std::min(n, precision + exp));
        out += "0";
        frac.append((size_t)zeros, '0');
        frac += d.substr(0, (size_t)t);
    }
    frac.append((size_t)(precision - (int)frac.size()), '0');
    return out + "." + frac;
}

}  // namespace kcfhost
