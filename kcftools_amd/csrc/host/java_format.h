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

// The same three renderings from the digits JDK 17 itself derives (java_fd17.cpp: a restatement of jdk.internal.math.
// FloatingDecimal, whose digits are not always the shortest -- JDK-4511638, fixed in JDK 19; kcftools targets Java 17,
// pom.xml:16). TEST INFRASTRUCTURE: the product prints from the shortest digits above; tests/test_host_side.py shows that
// the two agree on every value a KCF file can hold near a two-decimal tie.
std::string jdk17_fmt_2f(double x);
std::string jdk17_double_to_string(double x);
std::string jdk17_float_to_string(float x);
}  // namespace kcfhost
