// ic2/json.h -- minimal JSON DOM for the reference's .json model files (the reference uses picojson, which is not available
// to the product build).  Numbers are doubles, objects keep insertion order-independent lookup.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace snn {
namespace json {
class Value;
typedef std::vector<Value> Array;
typedef std::map<std::string, Value> Object;
class Value {
public:
    enum Type { Null, Bool, Number, String, ArrayT, ObjectT };
    Value() = default;
    Type type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Object> obj;
    bool isNumber() const { return type == Number; }
    bool isString() const { return type == String; }
    bool isArray() const { return type == ArrayT; }
    bool isObject() const { return type == ObjectT; }
    double asNumber() const {
        if (type != Number) throw std::runtime_error("json: not a number");
        return num;
    }
    const std::string& asString() const {
        if (type != String) throw std::runtime_error("json: not a string");
        return str;
    }
    const Array& asArray() const {
        if (type != ArrayT) throw std::runtime_error("json: not an array");
        return *arr;
    }
    const Object& asObject() const {
        if (type != ObjectT) throw std::runtime_error("json: not an object");
        return *obj;
    }
    bool has(const std::string& k) const { return type == ObjectT && obj->count(k) > 0; }
    const Value& at(const std::string& k) const {
        if (type != ObjectT) throw std::runtime_error("json: not an object (key " + k + ")");
        auto it = obj->find(k);
        if (it == obj->end()) throw std::runtime_error("json: missing key " + k);
        return it->second;
    }
};
// returns empty string on success, else an error message
std::string parse(Value& out, const std::string& text);
} // namespace json
} // namespace snn
