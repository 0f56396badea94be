// json.hpp -- the few lines of JSON the test/embedding harness (extension.cpp:
// pdk_query) needs to receive a call and hand back a result table.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pdkjson {

struct Json {
	enum Kind { NUL, BOOL, INT, DBL, STR, ARR, OBJ } kind = NUL;
	bool b = false;
	long long i = 0;
	double d = 0.0;
	std::string s;
	std::vector<Json> arr;
	std::vector<std::pair<std::string, Json>> obj; // insertion order kept

	const Json *Get(const std::string &key) const {
		for (auto &kv : obj) {
			if (kv.first == key) {
				return &kv.second;
			}
		}
		return nullptr;
	}
};

class Parser {
public:
	explicit Parser(const std::string &text) : t_(text) {
	}
	Json Parse() {
		Json j = Value();
		Ws();
		if (p_ != t_.size()) {
			Fail("trailing characters");
		}
		return j;
	}

private:
	[[noreturn]] void Fail(const char *what) {
		throw std::runtime_error(std::string("JSON: ") + what + " at offset " + std::to_string(p_));
	}
	void Ws() {
		while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) {
			p_++;
		}
	}
	bool Lit(const char *w) {
		size_t n = std::char_traits<char>::length(w);
		if (t_.compare(p_, n, w) == 0) {
			p_ += n;
			return true;
		}
		return false;
	}
	Json Value() {
		Ws();
		if (p_ >= t_.size()) {
			Fail("unexpected end");
		}
		Json j;
		char c = t_[p_];
		if (c == '{') {
			p_++;
			j.kind = Json::OBJ;
			Ws();
			if (t_[p_] == '}') {
				p_++;
				return j;
			}
			while (true) {
				Ws();
				Json k = Value();
				if (k.kind != Json::STR) {
					Fail("object key must be a string");
				}
				Ws();
				if (t_[p_++] != ':') {
					Fail("expected ':'");
				}
				j.obj.emplace_back(k.s, Value());
				Ws();
				if (t_[p_] == ',') {
					p_++;
					continue;
				}
				if (t_[p_] == '}') {
					p_++;
					return j;
				}
				Fail("expected ',' or '}'");
			}
		}
		if (c == '[') {
			p_++;
			j.kind = Json::ARR;
			Ws();
			if (t_[p_] == ']') {
				p_++;
				return j;
			}
			while (true) {
				j.arr.push_back(Value());
				Ws();
				if (t_[p_] == ',') {
					p_++;
					continue;
				}
				if (t_[p_] == ']') {
					p_++;
					return j;
				}
				Fail("expected ',' or ']'");
			}
		}
		if (c == '"') {
			p_++;
			j.kind = Json::STR;
			while (p_ < t_.size() && t_[p_] != '"') {
				if (t_[p_] == '\\' && p_ + 1 < t_.size()) {
					p_++;
					char e = t_[p_++];
					switch (e) {
					case 'n':
						j.s += '\n';
						break;
					case 't':
						j.s += '\t';
						break;
					case 'u':
						j.s += static_cast<char>(std::strtol(t_.substr(p_, 4).c_str(), nullptr, 16));
						p_ += 4;
						break;
					default:
						j.s += e;
					}
				} else {
					j.s += t_[p_++];
				}
			}
			p_++;
			return j;
		}
		if (Lit("null")) {
			return j;
		}
		if (Lit("true")) {
			j.kind = Json::BOOL;
			j.b = true;
			return j;
		}
		if (Lit("false")) {
			j.kind = Json::BOOL;
			return j;
		}
		size_t start = p_;
		bool is_dbl = false;
		while (p_ < t_.size() && (std::isdigit(static_cast<unsigned char>(t_[p_])) || t_[p_] == '-' || t_[p_] == '+' ||
		                          t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E')) {
			if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') {
				is_dbl = true;
			}
			p_++;
		}
		if (p_ == start) {
			Fail("unexpected character");
		}
		std::string num = t_.substr(start, p_ - start);
		if (is_dbl) {
			j.kind = Json::DBL;
			j.d = std::strtod(num.c_str(), nullptr);
		} else {
			j.kind = Json::INT;
			j.i = std::strtoll(num.c_str(), nullptr, 10);
		}
		return j;
	}
	const std::string &t_;
	size_t p_ = 0;
};

inline void EscapeTo(std::string &out, const std::string &s) {
	out += '"';
	for (char c : s) {
		switch (c) {
		case '"':
			out += "\\\"";
			break;
		case '\\':
			out += "\\\\";
			break;
		case '\n':
			out += "\\n";
			break;
		case '\t':
			out += "\\t";
			break;
		default:
			if (static_cast<unsigned char>(c) < 0x20) {
				char buf[8];
				std::snprintf(buf, sizeof buf, "\\u%04x", c);
				out += buf;
			} else {
				out += c;
			}
		}
	}
	out += '"';
}

inline void DoubleTo(std::string &out, double d) {
	if (std::isnan(d)) {
		out += "NaN";
	} else if (std::isinf(d)) {
		out += d > 0 ? "Infinity" : "-Infinity";
	} else {
		char buf[40];
		std::snprintf(buf, sizeof buf, "%.17g", d);
		out += buf;
	}
}

} // namespace pdkjson
