// java_format.h -- the three Java number renderings the KCF text format depends on (Locale.US, KCFTOOLS.java:31).
#pragma once
#include <string>

namespace kcfhost {
// String.format("%.2f", double): HALF_UP on the digits of the shortest repr (FormattedFloatingDecimal), e.g.
// 2.125 -> "2.13" and 1.005 -> "1.01" where C printf gives "2.12" / "1.00". Data.java:129-130, Window.java:143-148.
std::string java_fmt_2f(double x);
// String.valueOf(double) / Double.toString: KCFHeader.java:140-166 (weights in ##PARAM lines), e.g. 0.3 -> "0.3".
std::string java_double_to_string(double x);
// Float.toString(float): Window.java:151 concatenates the float meanVariations, e.g. 62 -> "62.0".
std::string java_float_to_string(float x);
}  // namespace kcfhost
