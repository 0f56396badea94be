// duck_api.hpp -- the slice of DuckDB's table-function API that PlinkingDuck's
// .pgen functions are written against, restated so the shells in this directory
// compile and run without DuckDB (its headers are not in this image).
//
// Names, argument meaning and call protocol follow DuckDB v1.5 as the reference
// uses it: TableFunction{bind, init_global, init_local,
// function}, GlobalTableFunctionState::MaxThreads(), DataChunk of at most
// STANDARD_VECTOR_SIZE rows, projection via TableFunctionInitInput::column_ids,
// errors as InvalidInputException / IOException / InternalException.  With
// DuckDB present a maintainer swaps this header for <duckdb.hpp>; INTEGRATION.md
// lists the handful of spots that differ.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace duckdb {

using idx_t = uint64_t;
using column_t = uint64_t;
using std::make_shared;
using std::shared_ptr;
using std::string;
using std::unique_ptr;
using std::vector;

template <class T, class... Args>
unique_ptr<T> make_uniq(Args &&...args) {
	return unique_ptr<T>(new T(std::forward<Args>(args)...));
}

constexpr idx_t STANDARD_VECTOR_SIZE = 2048;
constexpr column_t COLUMN_IDENTIFIER_ROW_ID = static_cast<column_t>(-1);

// ---- exceptions --------------------------------------------------------------
// printf-style constructors like DuckDB's (std::string arguments are accepted).

template <class T>
inline T PdkArg(T v) {
	return v;
}
inline const char *PdkArg(const string &s) {
	return s.c_str();
}

template <class... A>
inline string PdkFormat(const string &fmt, const A &...a) {
	if (sizeof...(A) == 0) {
		return fmt;
	}
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wformat-security"
#pragma GCC diagnostic ignored "-Wformat-nonliteral"
	int n = std::snprintf(nullptr, 0, fmt.c_str(), PdkArg(a)...);
	string out(static_cast<size_t>(n > 0 ? n : 0), '\0');
	if (n > 0) {
		std::snprintf(&out[0], static_cast<size_t>(n) + 1, fmt.c_str(), PdkArg(a)...);
	}
#pragma GCC diagnostic pop
	return out;
}

#define PDK_EXCEPTION(NAME, KIND)                                                                                      \
	class NAME : public std::runtime_error {                                                                           \
	public:                                                                                                            \
		template <class... A>                                                                                          \
		explicit NAME(const string &fmt, const A &...a) : std::runtime_error(PdkFormat(fmt, a...)) {                   \
		}                                                                                                              \
		static const char *Kind() {                                                                                    \
			return KIND;                                                                                               \
		}                                                                                                              \
	};
PDK_EXCEPTION(InvalidInputException, "Invalid Input Error")
PDK_EXCEPTION(IOException, "IO Error")
PDK_EXCEPTION(InternalException, "INTERNAL Error")
PDK_EXCEPTION(BinderException, "Binder Error")
#undef PDK_EXCEPTION

// ---- types ---------------------------------------------------------------------

enum class LogicalTypeId { SQLNULL, BOOLEAN, TINYINT, INTEGER, UINTEGER, BIGINT, DOUBLE, VARCHAR, LIST, ARRAY, STRUCT, ANY };

struct LogicalType;
using child_list_t = vector<std::pair<string, LogicalType>>;

struct LogicalType {
	LogicalTypeId id_ = LogicalTypeId::SQLNULL;
	shared_ptr<LogicalType> child;      // LIST / ARRAY element
	idx_t array_size = 0;               // ARRAY
	shared_ptr<child_list_t> fields;    // STRUCT

	LogicalType() = default;
	LogicalType(LogicalTypeId id) : id_(id) { // NOLINT: implicit like DuckDB's
	}
	LogicalTypeId id() const {
		return id_;
	}
	bool operator==(const LogicalType &o) const {
		return ToString() == o.ToString();
	}

	static constexpr LogicalTypeId SQLNULL = LogicalTypeId::SQLNULL;
	static constexpr LogicalTypeId BOOLEAN = LogicalTypeId::BOOLEAN;
	static constexpr LogicalTypeId TINYINT = LogicalTypeId::TINYINT;
	static constexpr LogicalTypeId INTEGER = LogicalTypeId::INTEGER;
	static constexpr LogicalTypeId UINTEGER = LogicalTypeId::UINTEGER;
	static constexpr LogicalTypeId BIGINT = LogicalTypeId::BIGINT;
	static constexpr LogicalTypeId DOUBLE = LogicalTypeId::DOUBLE;
	static constexpr LogicalTypeId VARCHAR = LogicalTypeId::VARCHAR;
	static constexpr LogicalTypeId ANY = LogicalTypeId::ANY;

	static LogicalType LIST(const LogicalType &elem) {
		LogicalType t(LogicalTypeId::LIST);
		t.child = make_shared<LogicalType>(elem);
		return t;
	}
	static LogicalType ARRAY(const LogicalType &elem, idx_t size) {
		LogicalType t(LogicalTypeId::ARRAY);
		t.child = make_shared<LogicalType>(elem);
		t.array_size = size;
		return t;
	}
	static LogicalType STRUCT(child_list_t children) {
		LogicalType t(LogicalTypeId::STRUCT);
		t.fields = make_shared<child_list_t>(std::move(children));
		return t;
	}

	idx_t FixedWidth() const {
		switch (id_) {
		case LogicalTypeId::BOOLEAN:
		case LogicalTypeId::TINYINT:
			return 1;
		case LogicalTypeId::INTEGER:
		case LogicalTypeId::UINTEGER:
			return 4;
		case LogicalTypeId::BIGINT:
		case LogicalTypeId::DOUBLE:
			return 8;
		default:
			return 0;
		}
	}

	// the spelling typeof() prints
	string ToString() const {
		switch (id_) {
		case LogicalTypeId::SQLNULL:
			return "\"NULL\"";
		case LogicalTypeId::BOOLEAN:
			return "BOOLEAN";
		case LogicalTypeId::TINYINT:
			return "TINYINT";
		case LogicalTypeId::INTEGER:
			return "INTEGER";
		case LogicalTypeId::UINTEGER:
			return "UINTEGER";
		case LogicalTypeId::BIGINT:
			return "BIGINT";
		case LogicalTypeId::DOUBLE:
			return "DOUBLE";
		case LogicalTypeId::VARCHAR:
			return "VARCHAR";
		case LogicalTypeId::ANY:
			return "ANY";
		case LogicalTypeId::LIST:
			return child->ToString() + "[]";
		case LogicalTypeId::ARRAY:
			return child->ToString() + "[" + std::to_string(array_size) + "]";
		case LogicalTypeId::STRUCT: {
			string s = "STRUCT(";
			for (size_t i = 0; i < fields->size(); i++) {
				s += (i ? ", " : "") + (*fields)[i].first + " " + (*fields)[i].second.ToString();
			}
			return s + ")";
		}
		}
		return "?";
	}
};

struct ListType {
	static const LogicalType &GetChildType(const LogicalType &t) {
		return *t.child;
	}
};
struct ArrayType {
	// DuckDB's ArrayType::MAX_ARRAY_SIZE
	static constexpr idx_t MAX_ARRAY_SIZE = 100000;
};
struct StructType {
	static const child_list_t &GetChildTypes(const LogicalType &t) {
		return *t.fields;
	}
};

// ---- Value (bind-time parameters) ---------------------------------------------

class Value {
public:
	Value() = default;
	explicit Value(const LogicalType &t) : type_(t) {
	}
	static Value BOOLEAN(bool b) {
		Value v(LogicalType::BOOLEAN);
		v.null_ = false;
		v.i_ = b;
		return v;
	}
	static Value BIGINT(int64_t x) {
		Value v(LogicalType::BIGINT);
		v.null_ = false;
		v.i_ = x;
		return v;
	}
	static Value INTEGER(int32_t x) {
		Value v(LogicalType::INTEGER);
		v.null_ = false;
		v.i_ = x;
		return v;
	}
	static Value DOUBLE(double x) {
		Value v(LogicalType::DOUBLE);
		v.null_ = false;
		v.d_ = x;
		return v;
	}
	static Value VARCHAR(const string &s) {
		Value v(LogicalType::VARCHAR);
		v.null_ = false;
		v.s_ = s;
		return v;
	}
	static Value LIST(const LogicalType &child_type, vector<Value> children) {
		Value v(LogicalType::LIST(child_type));
		v.null_ = false;
		v.children_ = std::move(children);
		return v;
	}
	static Value STRUCT(vector<std::pair<string, Value>> fields) {
		child_list_t types;
		Value v;
		for (auto &f : fields) {
			types.emplace_back(f.first, f.second.type());
			v.children_.push_back(std::move(f.second));
		}
		v.type_ = LogicalType::STRUCT(std::move(types));
		v.null_ = false;
		return v;
	}

	const LogicalType &type() const {
		return type_;
	}
	bool IsNull() const {
		return null_;
	}
	template <class T>
	T GetValue() const;
	const vector<Value> &Children() const {
		return children_;
	}

private:
	LogicalType type_;
	bool null_ = true;
	int64_t i_ = 0;
	double d_ = 0.0;
	string s_;
	vector<Value> children_;
};

template <>
inline string Value::GetValue<string>() const {
	switch (type_.id()) {
	case LogicalTypeId::VARCHAR:
		return s_;
	case LogicalTypeId::DOUBLE: {
		char buf[64];
		std::snprintf(buf, sizeof buf, "%g", d_);
		return buf;
	}
	default:
		return std::to_string(i_);
	}
}
template <>
inline bool Value::GetValue<bool>() const {
	return type_.id() == LogicalTypeId::DOUBLE ? d_ != 0.0 : i_ != 0;
}
template <>
inline int64_t Value::GetValue<int64_t>() const {
	return type_.id() == LogicalTypeId::DOUBLE ? static_cast<int64_t>(d_) : i_;
}
template <>
inline int32_t Value::GetValue<int32_t>() const {
	return static_cast<int32_t>(GetValue<int64_t>());
}
template <>
inline double Value::GetValue<double>() const {
	return type_.id() == LogicalTypeId::DOUBLE ? d_ : static_cast<double>(i_);
}

struct ListValue {
	static const vector<Value> &GetChildren(const Value &v) {
		return v.Children();
	}
};
struct StructValue {
	static const vector<Value> &GetChildren(const Value &v) {
		return v.Children();
	}
};

// ---- vectors / chunks ------------------------------------------------------------

struct list_entry_t {
	uint64_t offset = 0;
	uint64_t length = 0;
};

// Index of a string inside its vector's heap (DuckDB's string_t points into a
// StringHeap; an index keeps the same call shape).
struct string_t {
	uint32_t index = 0;
};

class ValidityMask {
public:
	void Reset(idx_t n) {
		bits_.assign((n + 63) / 64, ~0ull);
		all_valid_ = true;
	}
	void EnsureCapacity(idx_t n) {
		if (bits_.size() * 64 < n) {
			bits_.resize((n + 63) / 64, ~0ull);
		}
	}
	void SetInvalid(idx_t i) {
		EnsureCapacity(i + 1);
		bits_[i >> 6] &= ~(1ull << (i & 63));
		all_valid_ = false;
	}
	void SetValid(idx_t i) {
		EnsureCapacity(i + 1);
		bits_[i >> 6] |= 1ull << (i & 63);
	}
	bool RowIsValid(idx_t i) const {
		return i >= bits_.size() * 64 || ((bits_[i >> 6] >> (i & 63)) & 1ull);
	}
	uint64_t *GetData() {
		return bits_.data();
	}
	idx_t Capacity() const {
		return bits_.size() * 64;
	}

private:
	vector<uint64_t> bits_;
	bool all_valid_ = true;
};

class Vector {
public:
	Vector() = default;
	explicit Vector(const LogicalType &t, idx_t capacity = STANDARD_VECTOR_SIZE) {
		Initialize(t, capacity);
	}
	void Initialize(const LogicalType &t, idx_t capacity) {
		type = t;
		validity.Reset(capacity);
		heap.clear();
		list_size = 0;
		const idx_t w = t.FixedWidth();
		if (w) {
			data.assign(capacity * w, 0);
		} else if (t.id() == LogicalTypeId::VARCHAR) {
			data.assign(capacity * sizeof(string_t), 0);
		} else if (t.id() == LogicalTypeId::LIST) {
			data.assign(capacity * sizeof(list_entry_t), 0);
			children.clear();
			children.emplace_back(make_uniq<Vector>(*t.child, 0));
		} else if (t.id() == LogicalTypeId::ARRAY) {
			children.clear();
			children.emplace_back(make_uniq<Vector>(*t.child, capacity * t.array_size));
		} else if (t.id() == LogicalTypeId::STRUCT) {
			children.clear();
			for (auto &f : *t.fields) {
				children.emplace_back(make_uniq<Vector>(f.second, capacity));
			}
		}
	}
	void ReserveChild(idx_t n) { // LIST child growth
		children[0]->GrowTo(n);
	}
	//! DataChunk::Reset: every row valid again, no strings, empty lists -- buffers (a LIST child's included)
	//! are kept and not cleared, as DuckDB keeps a chunk's buffers across scan calls.
	void ResetKeepingBuffers() {
		validity.Reset(validity.Capacity());
		heap.clear();
		list_size = 0;
		for (auto &ch : children) {
			ch->ResetKeepingBuffers();
		}
	}

	//! Store one (possibly nested) Value at row i.
	void SetValue(idx_t i, const Value &v) {
		if (v.IsNull()) {
			validity.SetInvalid(i);
			return;
		}
		switch (type.id()) {
		case LogicalTypeId::BOOLEAN:
		case LogicalTypeId::TINYINT:
			reinterpret_cast<int8_t *>(data.data())[i] = static_cast<int8_t>(v.GetValue<int64_t>());
			break;
		case LogicalTypeId::INTEGER:
		case LogicalTypeId::UINTEGER:
			reinterpret_cast<int32_t *>(data.data())[i] = static_cast<int32_t>(v.GetValue<int64_t>());
			break;
		case LogicalTypeId::BIGINT:
			reinterpret_cast<int64_t *>(data.data())[i] = v.GetValue<int64_t>();
			break;
		case LogicalTypeId::DOUBLE:
			reinterpret_cast<double *>(data.data())[i] = v.GetValue<double>();
			break;
		case LogicalTypeId::VARCHAR:
			heap.push_back(v.GetValue<string>());
			reinterpret_cast<string_t *>(data.data())[i] = string_t {static_cast<uint32_t>(heap.size() - 1)};
			break;
		case LogicalTypeId::LIST: {
			auto &kids = v.Children();
			auto *entries = reinterpret_cast<list_entry_t *>(data.data());
			entries[i].offset = list_size;
			entries[i].length = kids.size();
			GrowListChild(list_size + kids.size());
			for (size_t k = 0; k < kids.size(); k++) {
				children[0]->SetValue(list_size + k, kids[k]);
			}
			list_size += kids.size();
			break;
		}
		case LogicalTypeId::STRUCT: {
			auto &kids = v.Children();
			for (size_t k = 0; k < kids.size() && k < children.size(); k++) {
				children[k]->SetValue(i, kids[k]);
			}
			break;
		}
		default:
			break;
		}
	}
	void GrowListChild(idx_t n) {
		Vector &c = *children[0];
		c.GrowTo(n);
	}
	void GrowTo(idx_t n) {
		const idx_t w = type.FixedWidth();
		if (w && data.size() < n * w) {
			data.resize(n * w, 0);
		} else if (type.id() == LogicalTypeId::VARCHAR && data.size() < n * sizeof(string_t)) {
			data.resize(n * sizeof(string_t), 0);
		} else if (type.id() == LogicalTypeId::LIST && data.size() < n * sizeof(list_entry_t)) {
			data.resize(n * sizeof(list_entry_t), 0);
		} else if (type.id() == LogicalTypeId::STRUCT) {
			for (auto &ch : children) {
				ch->GrowTo(n);
			}
		} else if (type.id() == LogicalTypeId::ARRAY) {
			children[0]->GrowTo(n * type.array_size);
		}
		validity.EnsureCapacity(n);
	}

	LogicalType type;
	vector<uint8_t> data;
	ValidityMask validity;
	vector<string> heap;                 // VARCHAR payloads
	vector<unique_ptr<Vector>> children; // LIST/ARRAY: [0] = element vector; STRUCT: fields
	idx_t list_size = 0;
};

struct FlatVector {
	template <class T>
	static T *GetData(Vector &v) {
		return reinterpret_cast<T *>(v.data.data());
	}
	static void SetNull(Vector &v, idx_t i, bool is_null) {
		if (is_null) {
			v.validity.SetInvalid(i);
		} else {
			v.validity.SetValid(i);
		}
	}
	static ValidityMask &Validity(Vector &v) {
		return v.validity;
	}
};
struct StringVector {
	static string_t AddString(Vector &v, const string &s) {
		v.heap.push_back(s);
		return string_t {static_cast<uint32_t>(v.heap.size() - 1)};
	}
};
struct ListVector {
	static idx_t GetListSize(const Vector &v) {
		return v.list_size;
	}
	static void SetListSize(Vector &v, idx_t n) {
		v.list_size = n;
	}
	static void Reserve(Vector &v, idx_t n) {
		v.ReserveChild(n);
	}
	static Vector &GetEntry(Vector &v) {
		return *v.children[0];
	}
};
struct ArrayVector {
	static Vector &GetEntry(Vector &v) {
		return *v.children[0];
	}
};
struct StructVector {
	static vector<unique_ptr<Vector>> &GetEntries(Vector &v) {
		return v.children;
	}
};

class DataChunk {
public:
	void Initialize(const vector<LogicalType> &types) {
		data.clear();
		for (auto &t : types) {
			data.emplace_back(t, STANDARD_VECTOR_SIZE);
		}
		count_ = 0;
	}
	void Reset() {
		for (auto &v : data) {
			v.ResetKeepingBuffers();
		}
		count_ = 0;
	}
	void SetCardinality(idx_t n) {
		count_ = n;
	}
	idx_t size() const {
		return count_;
	}
	vector<Vector> data;

private:
	idx_t count_ = 0;
};

inline void CompatSetOutputCardinality(DataChunk &output, idx_t n) {
	output.SetCardinality(n);
}

// ---- function plumbing --------------------------------------------------------------

struct FunctionData {
	virtual ~FunctionData() = default;
	template <class T>
	const T &Cast() const {
		return *static_cast<const T *>(this);
	}
};
using TableFunctionData = FunctionData;

struct GlobalTableFunctionState {
	virtual ~GlobalTableFunctionState() = default;
	virtual idx_t MaxThreads() const {
		return 1;
	}
	template <class T>
	T &Cast() {
		return *static_cast<T *>(this);
	}
};
struct LocalTableFunctionState {
	virtual ~LocalTableFunctionState() = default;
	template <class T>
	T &Cast() {
		return *static_cast<T *>(this);
	}
};

// The settings the shells read (DuckDB: ClientContext::TryGetCurrentSetting + TaskScheduler).
class ClientContext {
public:
	std::map<string, Value> settings;
	idx_t db_threads = 4;
	bool TryGetCurrentSetting(const string &name, Value &out) const {
		auto it = settings.find(name);
		if (it == settings.end()) {
			return false;
		}
		out = it->second;
		return true;
	}
};
struct ExecutionContext {
	ClientContext &client;
};

struct TableFunctionBindInput {
	vector<Value> inputs;
	std::map<string, Value> named_parameters;
};
struct TableFunctionInitInput {
	const FunctionData *bind_data = nullptr;
	vector<column_t> column_ids;
};
struct TableFunctionInput {
	const FunctionData *bind_data = nullptr;
	GlobalTableFunctionState *global_state = nullptr;
	LocalTableFunctionState *local_state = nullptr;
};

using table_function_bind_t = unique_ptr<FunctionData> (*)(ClientContext &, TableFunctionBindInput &,
                                                           vector<LogicalType> &, vector<string> &);
using table_function_init_global_t = unique_ptr<GlobalTableFunctionState> (*)(ClientContext &,
                                                                              TableFunctionInitInput &);
using table_function_init_local_t = unique_ptr<LocalTableFunctionState> (*)(ExecutionContext &,
                                                                            TableFunctionInitInput &,
                                                                            GlobalTableFunctionState *);
using table_function_t = void (*)(ClientContext &, TableFunctionInput &, DataChunk &);

class TableFunction {
public:
	TableFunction() = default;
	TableFunction(string name_p, vector<LogicalType> arguments_p, table_function_t function_p,
	              table_function_bind_t bind_p, table_function_init_global_t init_global_p,
	              table_function_init_local_t init_local_p)
	    : name(std::move(name_p)), arguments(std::move(arguments_p)), function(function_p), bind(bind_p),
	      init_global(init_global_p), init_local(init_local_p) {
	}
	string name;
	vector<LogicalType> arguments;
	table_function_t function = nullptr;
	table_function_bind_t bind = nullptr;
	table_function_init_global_t init_global = nullptr;
	table_function_init_local_t init_local = nullptr;
	bool projection_pushdown = false;
	std::map<string, LogicalType> named_parameters;
};

class ExtensionLoader {
public:
	void RegisterFunction(const TableFunction &f) {
		functions[f.name] = f;
	}
	std::map<string, TableFunction> functions;
};

} // namespace duckdb
