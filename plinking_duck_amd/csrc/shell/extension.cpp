// extension.cpp -- registration of the six table functions (the reference's
// LoadInternal, src/plinking_duck_extension.cpp:89-102) and a C entry point that
// drives one of them through DuckDB's call protocol without DuckDB:
//
//   bind -> init_global -> MaxThreads() -> per thread { init_local; scan until 0 rows }
//
// pdk_query() is what the tests and embedders call; with DuckDB present the same
// Register* functions are handed to its ExtensionLoader instead.
#include "duck_api.hpp"
#include "json.hpp"
#include "plink_common.hpp"

#include <algorithm>
#include <mutex>
#include <chrono>
#include <thread>

namespace duckdb {

void RegisterPgenReader(ExtensionLoader &loader);
void RegisterPlinkFreq(ExtensionLoader &loader);
void RegisterPlinkHardy(ExtensionLoader &loader);
void RegisterPlinkMissing(ExtensionLoader &loader);
void RegisterPlinkScore(ExtensionLoader &loader);
void RegisterPlinkPca(ExtensionLoader &loader);
void RegisterPlinkLd(ExtensionLoader &loader);
void RegisterPfileReader(ExtensionLoader &loader);

static void LoadInternal(ExtensionLoader &loader) {
	RegisterPgenReader(loader);
	RegisterPlinkFreq(loader);
	RegisterPlinkHardy(loader);
	RegisterPlinkMissing(loader);
	RegisterPlinkScore(loader);
	RegisterPlinkPca(loader);
	RegisterPlinkLd(loader);
	RegisterPfileReader(loader);
}

namespace {

using pdkjson::Json;

Value JsonToValue(const Json &j) {
	switch (j.kind) {
	case Json::NUL:
		return Value();
	case Json::BOOL:
		return Value::BOOLEAN(j.b);
	case Json::INT:
		return Value::BIGINT(j.i);
	case Json::DBL:
		return Value::DOUBLE(j.d);
	case Json::STR:
		return Value::VARCHAR(j.s);
	case Json::ARR: {
		vector<Value> kids;
		for (auto &e : j.arr) {
			kids.push_back(JsonToValue(e));
		}
		LogicalType child = LogicalType::INTEGER;
		bool any_double = false, all_numeric = !kids.empty();
		for (auto &k : kids) {
			auto id = k.type().id();
			any_double |= id == LogicalTypeId::DOUBLE;
			all_numeric &= id == LogicalTypeId::DOUBLE || id == LogicalTypeId::BIGINT;
		}
		if (!kids.empty()) {
			child = (all_numeric && any_double) ? LogicalType(LogicalType::DOUBLE) : kids[0].type();
		}
		return Value::LIST(child, std::move(kids));
	}
	case Json::OBJ: {
		vector<std::pair<string, Value>> fields;
		for (auto &kv : j.obj) {
			fields.emplace_back(kv.first, JsonToValue(kv.second));
		}
		return Value::STRUCT(std::move(fields));
	}
	}
	return Value();
}

void CellToJson(string &out, Vector &v, idx_t row) {
	if (!v.validity.RowIsValid(row)) {
		out += "null";
		return;
	}
	switch (v.type.id()) {
	case LogicalTypeId::BOOLEAN:
		out += FlatVector::GetData<int8_t>(v)[row] ? "true" : "false";
		break;
	case LogicalTypeId::TINYINT:
		out += std::to_string(static_cast<int>(FlatVector::GetData<int8_t>(v)[row]));
		break;
	case LogicalTypeId::INTEGER:
		out += std::to_string(FlatVector::GetData<int32_t>(v)[row]);
		break;
	case LogicalTypeId::UINTEGER:
		out += std::to_string(FlatVector::GetData<uint32_t>(v)[row]);
		break;
	case LogicalTypeId::BIGINT:
		out += std::to_string(FlatVector::GetData<int64_t>(v)[row]);
		break;
	case LogicalTypeId::DOUBLE:
		pdkjson::DoubleTo(out, FlatVector::GetData<double>(v)[row]);
		break;
	case LogicalTypeId::VARCHAR:
		pdkjson::EscapeTo(out, v.heap[FlatVector::GetData<string_t>(v)[row].index]);
		break;
	case LogicalTypeId::LIST: {
		auto e = FlatVector::GetData<list_entry_t>(v)[row];
		out += '[';
		for (idx_t k = 0; k < e.length; k++) {
			if (k) {
				out += ',';
			}
			CellToJson(out, *v.children[0], e.offset + k);
		}
		out += ']';
		break;
	}
	case LogicalTypeId::ARRAY: {
		out += '[';
		for (idx_t k = 0; k < v.type.array_size; k++) {
			if (k) {
				out += ',';
			}
			CellToJson(out, *v.children[0], row * v.type.array_size + k);
		}
		out += ']';
		break;
	}
	case LogicalTypeId::STRUCT: {
		out += '{';
		for (size_t k = 0; k < v.children.size(); k++) {
			if (k) {
				out += ',';
			}
			pdkjson::EscapeTo(out, (*v.type.fields)[k].first);
			out += ':';
			CellToJson(out, *v.children[k], row);
		}
		out += '}';
		break;
	}
	default:
		out += "null";
	}
}

//! Reads every valid cell of the first `rows` rows once (order-independent sum): the consumer's side of a chunk.
uint64_t DrainColumn(Vector &v, idx_t rows) {
	uint64_t sum = 0;
	auto bits = [](double d) {
		uint64_t u;
		std::memcpy(&u, &d, sizeof u);
		return u;
	};
	switch (v.type.id()) {
	case LogicalTypeId::BOOLEAN:
	case LogicalTypeId::TINYINT: {
		// a genotype child holds up to 1e9 cells per chunk: bytes and validity words are summed apart
		// (the shells store 0 under an invalid cell)
		const uint8_t *p = reinterpret_cast<const uint8_t *>(FlatVector::GetData<int8_t>(v));
		uint64_t acc[4] = {0, 0, 0, 0};
		idx_t r = 0;
		for (; r + 32 <= rows; r += 32) {
			for (int k = 0; k < 4; k++) {
				uint64_t w;
				std::memcpy(&w, p + r + 8 * k, 8);
				acc[k] += (w & 0x00ff00ff00ff00ffull) + ((w >> 8) & 0x00ff00ff00ff00ffull);
			}
			if ((r & 0xfff) == 0xfe0) { // fold the 16-bit lanes before they can overflow
				for (int k = 0; k < 4; k++) {
					sum += (acc[k] & 0xffff) + ((acc[k] >> 16) & 0xffff) + ((acc[k] >> 32) & 0xffff) + (acc[k] >> 48);
					acc[k] = 0;
				}
			}
		}
		for (int k = 0; k < 4; k++) {
			sum += (acc[k] & 0xffff) + ((acc[k] >> 16) & 0xffff) + ((acc[k] >> 32) & 0xffff) + (acc[k] >> 48);
		}
		for (; r < rows; r++) {
			sum += p[r];
		}
		const uint64_t *bitsw = v.validity.GetData();
		for (idx_t w = 0; w < std::min(rows, v.validity.Capacity()) / 64; w++) {
			sum += 0x9eull * static_cast<uint64_t>(64 - __builtin_popcountll(bitsw[w]));
		}
		for (idx_t i = std::min(rows, v.validity.Capacity()) / 64 * 64; i < rows; i++) {
			sum += v.validity.RowIsValid(i) ? 0 : 0x9eu;
		}
		break;
	}
	case LogicalTypeId::INTEGER:
	case LogicalTypeId::UINTEGER:
		for (idx_t r = 0; r < rows; r++) {
			sum += v.validity.RowIsValid(r) ? FlatVector::GetData<uint32_t>(v)[r] : 0x9e37u;
		}
		break;
	case LogicalTypeId::BIGINT:
		for (idx_t r = 0; r < rows; r++) {
			sum += v.validity.RowIsValid(r) ? static_cast<uint64_t>(FlatVector::GetData<int64_t>(v)[r]) : 0x9e37u;
		}
		break;
	case LogicalTypeId::DOUBLE:
		for (idx_t r = 0; r < rows; r++) {
			sum += v.validity.RowIsValid(r) ? bits(FlatVector::GetData<double>(v)[r]) : 0x9e37u;
		}
		break;
	case LogicalTypeId::VARCHAR:
		for (idx_t r = 0; r < rows; r++) {
			if (v.validity.RowIsValid(r)) {
				const string &str = v.heap[FlatVector::GetData<string_t>(v)[r].index];
				sum += str.size() + (str.empty() ? 0u : static_cast<uint8_t>(str.back()));
			}
		}
		break;
	case LogicalTypeId::LIST:
		for (idx_t r = 0; r < rows; r++) {
			if (v.validity.RowIsValid(r)) {
				sum += FlatVector::GetData<list_entry_t>(v)[r].length;
			}
		}
		if (rows) {
			const auto last = FlatVector::GetData<list_entry_t>(v)[rows - 1];
			sum += DrainColumn(*v.children[0], last.offset + last.length);
		}
		break;
	case LogicalTypeId::ARRAY:
		sum += DrainColumn(*v.children[0], rows * v.type.array_size);
		break;
	case LogicalTypeId::STRUCT:
		for (auto &child : v.children) {
			sum += DrainColumn(*child, rows);
		}
		break;
	default:
		break;
	}
	return sum;
}

string ErrorJson(const char *kind, const string &msg) {
	string out = "{\"error\":{\"kind\":";
	pdkjson::EscapeTo(out, kind);
	out += ",\"message\":";
	pdkjson::EscapeTo(out, msg);
	out += "}}";
	return out;
}

string RunQuery(const string &request) {
	Json req = pdkjson::Parser(request).Parse();
	ExtensionLoader loader;
	LoadInternal(loader);
	const Json *fn = req.Get("function");
	if (!fn || fn->kind != Json::STR) {
		return ErrorJson("Binder Error", "request needs a \"function\" name");
	}
	auto it = loader.functions.find(fn->s);
	if (it == loader.functions.end()) {
		return ErrorJson("Catalog Error", "Table Function with name " + fn->s + " does not exist!");
	}
	TableFunction &tf = it->second;

	ClientContext context;
	if (const Json *th = req.Get("threads")) {
		context.db_threads = static_cast<idx_t>(std::max<long long>(1, th->i));
	}
	if (const Json *st = req.Get("settings")) {
		for (auto &kv : st->obj) {
			Value v = JsonToValue(kv.second);
			// the SET-time check of the option (src/plinking_duck_extension.cpp:37-42)
			if (kv.first == "plinking_max_threads" && !v.IsNull() && v.GetValue<int64_t>() < 0) {
				return ErrorJson(InvalidInputException::Kind(),
				                 "plinking_max_threads must be non-negative (0 = default, >0 = cap)");
			}
			if (kv.first == "plinking_devices") {
				// the option's SET callback: which GPUs hold the variant shards of files opened from now on
				try {
					SetPlinkingDevices(v.IsNull() ? string() : v.GetValue<string>());
				} catch (const InvalidInputException &ex) {
					return ErrorJson(InvalidInputException::Kind(), ex.what());
				}
			}
			context.settings[kv.first] = v;
		}
	}
	TableFunctionBindInput bind_input;
	if (const Json *args = req.Get("args")) {
		for (auto &a : args->arr) {
			bind_input.inputs.push_back(JsonToValue(a));
		}
	}
	if (bind_input.inputs.size() != tf.arguments.size()) {
		return ErrorJson("Binder Error", "No function matches the given name and argument types '" + tf.name + "()'");
	}
	if (const Json *named = req.Get("named")) {
		for (auto &kv : named->obj) {
			if (!tf.named_parameters.count(kv.first)) {
				return ErrorJson("Binder Error",
				                 "Invalid named parameter \"" + kv.first + "\" for function " + tf.name);
			}
			bind_input.named_parameters[kv.first] = JsonToValue(kv.second);
		}
	}

	vector<LogicalType> return_types;
	vector<string> names;
	const auto t_bind0 = std::chrono::steady_clock::now();
	auto bind_data = tf.bind(context, bind_input, return_types, names);
	const auto t_bind1 = std::chrono::steady_clock::now();

	// projection pushdown: "columns": [names] or absent for all
	TableFunctionInitInput init_input;
	init_input.bind_data = bind_data.get();
	vector<idx_t> projected;
	if (const Json *cols = req.Get("columns")) {
		for (auto &c : cols->arr) {
			auto pos = std::find(names.begin(), names.end(), c.s);
			if (pos == names.end()) {
				return ErrorJson("Binder Error", "Referenced column \"" + c.s + "\" not found in FROM clause!");
			}
			projected.push_back(static_cast<idx_t>(pos - names.begin()));
		}
	} else {
		for (idx_t i = 0; i < names.size(); i++) {
			projected.push_back(i);
		}
	}
	// A function without projection_pushdown always fills every column; DuckDB projects above it.
	vector<idx_t> scanned = projected;
	if (!tf.projection_pushdown) {
		scanned.clear();
		for (idx_t i = 0; i < names.size(); i++) {
			scanned.push_back(i);
		}
	}
	vector<idx_t> emit; // positions inside the scanned chunk that the caller asked for
	for (auto c : projected) {
		emit.push_back(static_cast<idx_t>(std::find(scanned.begin(), scanned.end(), c) - scanned.begin()));
	}
	init_input.column_ids.assign(scanned.begin(), scanned.end());
	vector<LogicalType> out_types;
	for (auto c : scanned) {
		out_types.push_back(return_types[c]);
	}

	const auto t_init0 = std::chrono::steady_clock::now();
	auto gstate = tf.init_global(context, init_input);
	const auto t_init1 = std::chrono::steady_clock::now();
	idx_t n_threads = std::max<idx_t>(1, std::min<idx_t>(gstate->MaxThreads(), context.db_threads));

	// "drain": true -- consume the chunks as DuckDB's pipeline would (every projected cell is read once) without
	// serialising rows for the caller: what tools/shell_bench.py times at full size, where a JSON copy of a million
	// rows would dwarf the scan.  The reply then carries row_count and a checksum instead of rows.
	bool drain = false;
	if (const Json *d = req.Get("drain")) {
		drain = d->kind == Json::BOOL && d->b;
	}
	std::mutex result_mutex;
	vector<string> row_chunks;
	idx_t total_rows = 0;
	uint64_t drain_checksum = 0;
	string first_error, first_error_kind;
	auto worker = [&]() {
		try {
			ExecutionContext exec {context};
			auto lstate = tf.init_local(exec, init_input, gstate.get());
			TableFunctionInput in;
			in.bind_data = bind_data.get();
			in.global_state = gstate.get();
			in.local_state = lstate.get();
			DataChunk chunk;
			chunk.Initialize(out_types);
			while (true) {
				chunk.Reset();
				tf.function(context, in, chunk);
				if (chunk.size() == 0) {
					break; // DuckDB marks the thread FINISHED on an empty chunk
				}
				if (drain) {
					uint64_t sum = 0;
					for (size_t c = 0; c < emit.size(); c++) {
						sum += DrainColumn(chunk.data[emit[c]], chunk.size());
					}
					std::lock_guard<std::mutex> lock(result_mutex);
					total_rows += chunk.size();
					drain_checksum += sum;
					continue;
				}
				string rows;
				for (idx_t r = 0; r < chunk.size(); r++) {
					rows += rows.empty() ? "[" : ",[";
					for (size_t c = 0; c < emit.size(); c++) {
						if (c) {
							rows += ',';
						}
						CellToJson(rows, chunk.data[emit[c]], r);
					}
					rows += ']';
				}
				std::lock_guard<std::mutex> lock(result_mutex);
				total_rows += chunk.size();
				row_chunks.push_back(std::move(rows));
			}
		} catch (const InvalidInputException &e) {
			std::lock_guard<std::mutex> lock(result_mutex);
			if (first_error.empty()) {
				first_error = e.what();
				first_error_kind = InvalidInputException::Kind();
			}
		} catch (const IOException &e) {
			std::lock_guard<std::mutex> lock(result_mutex);
			if (first_error.empty()) {
				first_error = e.what();
				first_error_kind = IOException::Kind();
			}
		} catch (const std::exception &e) {
			std::lock_guard<std::mutex> lock(result_mutex);
			if (first_error.empty()) {
				first_error = e.what();
				first_error_kind = InternalException::Kind();
			}
		}
	};
	vector<std::thread> threads;
	for (idx_t t = 1; t < n_threads; t++) {
		threads.emplace_back(worker);
	}
	worker();
	for (auto &t : threads) {
		t.join();
	}
	const auto t_scan1 = std::chrono::steady_clock::now();
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
		return std::chrono::duration<double, std::milli>(b - a).count();
	};
	if (!first_error.empty()) {
		return ErrorJson(first_error_kind.c_str(), first_error);
	}

	string out = "{\"names\":[";
	for (size_t i = 0; i < projected.size(); i++) {
		if (i) {
			out += ',';
		}
		pdkjson::EscapeTo(out, names[projected[i]]);
	}
	out += "],\"types\":[";
	for (size_t i = 0; i < projected.size(); i++) {
		if (i) {
			out += ',';
		}
		pdkjson::EscapeTo(out, out_types[i].ToString());
	}
	out += "],\"all_names\":[";
	for (size_t i = 0; i < names.size(); i++) {
		if (i) {
			out += ',';
		}
		pdkjson::EscapeTo(out, names[i]);
	}
	out += "],\"all_types\":[";
	for (size_t i = 0; i < return_types.size(); i++) {
		if (i) {
			out += ',';
		}
		pdkjson::EscapeTo(out, return_types[i].ToString());
	}
	// phase times of this call (bind = companions + header probe, init = device residency, scan = threads)
	out += "],\"bind_ms\":" + std::to_string(ms(t_bind0, t_bind1)) + ",\"init_ms\":" + std::to_string(ms(t_init0, t_init1)) +
	       ",\"scan_ms\":" + std::to_string(ms(t_init1, t_scan1));
	out += ",\"threads\":" + std::to_string(n_threads) + ",\"row_count\":" + std::to_string(total_rows);
	if (drain) {
		out += ",\"checksum\":" + std::to_string(drain_checksum);
	}
	out += ",\"rows\":[";
	bool first = true;
	for (auto &rc : row_chunks) {
		if (!first) {
			out += ',';
		}
		out += rc;
		first = false;
	}
	out += "]}";
	return out;
}

} // namespace
} // namespace duckdb

extern "C" {

// Run one table function call described in JSON; returns a malloc'ed JSON result
// (free with pdk_free).  Never throws: errors come back as {"error": {...}}.
char *pdk_query(const char *request_json) {
	std::string out;
	try {
		out = duckdb::RunQuery(request_json ? request_json : "");
	} catch (const duckdb::InvalidInputException &e) {
		out = duckdb::ErrorJson(duckdb::InvalidInputException::Kind(), e.what());
	} catch (const duckdb::IOException &e) {
		out = duckdb::ErrorJson(duckdb::IOException::Kind(), e.what());
	} catch (const std::exception &e) {
		out = duckdb::ErrorJson("Error", e.what());
	}
	char *buf = static_cast<char *>(std::malloc(out.size() + 1));
	std::memcpy(buf, out.c_str(), out.size() + 1);
	return buf;
}

void pdk_free(char *p) {
	std::free(p);
}

// Names of the registered table functions, comma separated (static storage).
const char *pdk_functions(void) {
	static std::string names;
	if (names.empty()) {
		duckdb::ExtensionLoader loader;
		duckdb::LoadInternal(loader);
		for (auto &kv : loader.functions) {
			names += (names.empty() ? "" : ",") + kv.first;
		}
	}
	return names.c_str();
}
}
