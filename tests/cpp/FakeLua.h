// FakeLua.h -- a tiny stand-in for a Garry's Mod Lua state, enough to drive the Tracing API
// thunks without the game: a value stack, ordered tables, typed userdata, and errors raised as
// C++ exceptions (the real ThrowError/ArgError/CheckType longjmp out of the C function).
#pragma once

#include <algorithm>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "LuaShim.h"

namespace fakelua {

using namespace GarrysMod::Lua;

struct LuaError : std::runtime_error {
    int arg;   // ArgError argument number, 0 otherwise
    LuaError(const std::string& m, int a = 0) : std::runtime_error(m), arg(a) {}
};

struct Table;
struct Value {
    int type = Type::Nil;
    double num = 0;
    bool b = false;
    std::string str;
    ::Vector vec{0, 0, 0};
    std::shared_ptr<Table> tab;
    std::shared_ptr<void*> ud;   // boxed pointer so SetUserType(…, NULL) is seen by all copies
    CFunc fn = nullptr;
};
struct Table { std::vector<std::pair<Value, Value>> kv; };

inline bool key_equal(const Value& a, const Value& b)
{
    if (a.type != b.type) return false;
    if (a.type == Type::Number) return a.num == b.num;
    if (a.type == Type::String) return a.str == b.str;
    return false;
}

class State : public ILuaBase {
public:
    std::vector<Value> stack;
    std::vector<std::string> typeNames;   // user types from CreateMetaTable: id = Type::Count + index
    std::vector<std::shared_ptr<Table>> metatables;   // same index as typeNames
    Value globals = NewTable();           // _G

    // obj:name(...) the way Lua resolves it: metatable of the value's type -> __index -> field
    const Value* find_method(int type, const char* name) const
    {
        if (type < Type::Count || size_t(type - Type::Count) >= metatables.size()) return nullptr;
        const Table& mt = *metatables[size_t(type - Type::Count)];
        const Table* index = nullptr;
        for (auto& kv : mt.kv)
            if (kv.first.type == Type::String && kv.first.str == "__index" && kv.second.type == Type::Table) index = kv.second.tab.get();
        if (!index) return nullptr;
        for (auto& kv : index->kv)
            if (kv.first.type == Type::String && kv.first.str == name) return &kv.second;
        return nullptr;
    }
    const Value* find_global(const char* table, const char* name) const
    {
        for (auto& kv : globals.tab->kv)
            if (kv.first.type == Type::String && kv.first.str == table && kv.second.type == Type::Table)
                for (auto& f : kv.second.tab->kv)
                    if (f.first.type == Type::String && f.first.str == name) return &f.second;
        return nullptr;
    }

    static Value Num(double d) { Value v; v.type = Type::Number; v.num = d; return v; }
    static Value Bool(bool b) { Value v; v.type = Type::Bool; v.b = b; return v; }
    static Value Str(const std::string& s) { Value v; v.type = Type::String; v.str = s; return v; }
    static Value Vec(float x, float y, float z) { Value v; v.type = Type::Vector; v.vec = ::Vector{x, y, z}; return v; }
    static Value Nil() { return Value(); }
    static Value NewTable() { Value v; v.type = Type::Table; v.tab = std::make_shared<Table>(); return v; }
    static Value User(void* p, int type) { Value v; v.type = type; v.ud = std::make_shared<void*>(p); return v; }
    static Value Array(const std::vector<Value>& items)
    {
        Value t = NewTable();
        for (size_t i = 0; i < items.size(); ++i) t.tab->kv.push_back({Num(double(i + 1)), items[i]});
        return t;
    }
    void PushValue(const Value& v) { stack.push_back(v); }

    int abs_index(int pos) const { return pos > 0 ? pos : int(stack.size()) + pos + 1; }
    Value* at(int pos)
    {
        int i = abs_index(pos);
        return (i >= 1 && i <= int(stack.size())) ? &stack[size_t(i - 1)] : nullptr;
    }
    std::string type_name(int t) const
    {
        static const char* names[] = {"nil", "boolean", "lightuserdata", "number", "string", "table", "function",
                                      "userdata", "thread", "Entity", "Vector", "Angle"};
        if (t == Type::None) return "no value";
        if (t >= 0 && t <= Type::Angle) return names[t];
        if (t >= Type::Count && size_t(t - Type::Count) < typeNames.size()) return typeNames[size_t(t - Type::Count)];
        return "userdata";
    }

    // ---- ILuaBase -----------------------------------------------------------------------------
    int Top() override { return int(stack.size()); }
    void Push(int pos) override { Value v = *at(pos); stack.push_back(v); }
    void Pop(int amount = 1) override
    {
        if (amount > int(stack.size())) throw std::logic_error("FakeLua: pop past the bottom of the stack");
        stack.resize(stack.size() - size_t(amount));
    }
    void CreateTable() override { stack.push_back(NewTable()); }
    void PushSpecial(int type) override
    {
        if (type != SPECIAL_GLOB) throw std::logic_error("FakeLua: only SPECIAL_GLOB is modelled");
        stack.push_back(globals);
    }
    void GetField(int pos, const char* name) override
    {
        Value* t = at(pos);
        if (!t || t->type != Type::Table) throw std::logic_error("FakeLua: GetField on a non-table");
        Value found = Nil();
        for (auto& kv : t->tab->kv)
            if (kv.first.type == Type::String && kv.first.str == name) found = kv.second;
        stack.push_back(found);
    }
    void GetTable(int pos) override
    {
        Value* t = at(pos);
        if (!t || t->type != Type::Table) throw std::logic_error("FakeLua: GetTable on a non-table");
        std::shared_ptr<Table> tab = t->tab;
        const Value key = stack.back();
        stack.pop_back();
        // array part: t[k] of a table filled in order 1, 2, 3, ... sits at kv[k - 1] (what a Lua VM's array part gives in
        // O(1); without it the batch benchmarks would measure this fake's linear search, not the binding)
        if (const Value* hit = array_slot(*tab, key)) { stack.push_back(*hit); return; }
        Value found = Nil();
        for (auto& kv : tab->kv)
            if (key_equal(kv.first, key)) found = kv.second;
        stack.push_back(found);
    }
    static Value* array_slot(Table& t, const Value& key)
    {
        if (key.type != Type::Number || !(key.num >= 1.0) || key.num > double(t.kv.size())) return nullptr;
        const size_t k = size_t(key.num);
        if (double(k) != key.num) return nullptr;
        auto& kv = t.kv[k - 1];
        return (kv.first.type == Type::Number && kv.first.num == key.num) ? &kv.second : nullptr;
    }
    void Call(int nargs, int nresults) override
    {
        if (int(stack.size()) < nargs + 1) throw std::logic_error("FakeLua: Call with too few values on the stack");
        const size_t base = stack.size() - size_t(nargs) - 1;
        const Value fn = stack[base];
        std::vector<Value> frame(stack.begin() + long(base) + 1, stack.end());
        stack.resize(base);
        if (fn.type != Type::Function || !fn.fn) throw LuaError("attempt to call a " + type_name(fn.type) + " value");
        frame.swap(stack);                         // the callee sees only its arguments
        int nret = 0;
        try { nret = fn.fn(this); } catch (...) { frame.swap(stack); throw; }
        std::vector<Value> rets(stack.end() - std::min<long>(nret, long(stack.size())), stack.end());
        frame.swap(stack);
        for (int i = 0; i < nresults; ++i) stack.push_back(i < int(rets.size()) ? rets[size_t(i)] : Nil());
    }
    void SetField(int pos, const char* name) override
    {
        Value val = stack.back();
        Value* t = at(pos);
        set(*t, Str(name), val);
        stack.pop_back();
    }
    void SetTable(int pos) override
    {
        Value val = stack.back(), key = stack[stack.size() - 2];
        Value* t = at(pos);
        set(*t, key, val);
        stack.pop_back();
        stack.pop_back();
    }
    void SetMetaTable(int) override { stack.pop_back(); }
    int Next(int pos) override
    {
        Value* t = at(pos);
        if (!t || t->type != Type::Table) throw std::logic_error("FakeLua: Next on a non-table");
        std::shared_ptr<Table> tab = t->tab;   // keep alive: popping may invalidate `t`
        Value key = stack.back();
        stack.pop_back();
        size_t next = 0;
        if (key.type != Type::Nil) {
            next = tab->kv.size();
            for (size_t i = 0; i < tab->kv.size(); ++i)
                if (key_equal(tab->kv[i].first, key)) { next = i + 1; break; }
        }
        if (next >= tab->kv.size()) return 0;
        stack.push_back(tab->kv[next].first);
        stack.push_back(tab->kv[next].second);
        return 1;
    }
    [[noreturn]] void ThrowError(const char* msg) override { throw LuaError(msg); }
    void CheckType(int pos, int type) override
    {
        int got = GetType(pos);
        if (got != type)
            throw LuaError("bad argument #" + std::to_string(pos) + " (" + type_name(type) + " expected, got " +
                               type_name(got) + ")", pos);
    }
    [[noreturn]] void ArgError(int argNum, const char* msg) override
    {
        throw LuaError("bad argument #" + std::to_string(argNum) + " (" + msg + ")", argNum);
    }
    bool IsType(int pos, int type) override { return GetType(pos) == type; }
    int GetType(int pos) override { Value* v = at(pos); return v ? v->type : int(Type::None); }
    double GetNumber(int pos = -1) override { Value* v = at(pos); return v && v->type == Type::Number ? v->num : 0.0; }
    double CheckNumber(int pos = -1) override { CheckType(pos, Type::Number); return at(pos)->num; }
    bool GetBool(int pos = -1) override { Value* v = at(pos); return v && v->type == Type::Bool && v->b; }
    const char* GetString(int pos = -1, unsigned int* outLen = nullptr) override
    {
        Value* v = at(pos);
        if (!v || v->type != Type::String) { if (outLen) *outLen = 0; return nullptr; }
        if (outLen) *outLen = unsigned(v->str.size());
        return v->str.data();
    }
    const ::Vector& GetVector(int pos = -1) override
    {
        static const ::Vector zero{0, 0, 0};
        Value* v = at(pos);
        return v && v->type == Type::Vector ? v->vec : zero;
    }
    void PushNil() override { stack.push_back(Nil()); }
    void PushNumber(double v) override { stack.push_back(Num(v)); }
    void PushBool(bool v) override { stack.push_back(Bool(v)); }
    void PushString(const char* s, unsigned len = 0) override
    {   // ONE copy of the bytes, as lua_pushlstring makes
        stack.emplace_back();
        stack.back().type = Type::String;
        if (len) stack.back().str.assign(s, len); else stack.back().str.assign(s);
    }
    void PushVector(const ::Vector& v) override { stack.push_back(Vec(v.x, v.y, v.z)); }
    void PushCFunction(CFunc f) override { Value v; v.type = Type::Function; v.fn = f; stack.push_back(v); }
    int CreateMetaTable(const char* name) override
    {
        typeNames.push_back(name);
        stack.push_back(NewTable());
        metatables.push_back(stack.back().tab);
        return Type::Count + int(typeNames.size()) - 1;
    }
    void PushUserType(void* data, int type) override { stack.push_back(User(data, type)); }
    void SetUserType(int pos, void* data) override { Value* v = at(pos); if (v && v->ud) *v->ud = data; }
    void* GetUserdataRaw(int pos, int type) override
    {
        Value* v = at(pos);
        return (v && v->type == type && v->ud) ? *v->ud : nullptr;
    }

private:
    static void set(Value& t, const Value& key, const Value& val)
    {
        if (t.type != Type::Table) throw std::logic_error("FakeLua: field set on a non-table");
        if (Value* slot = array_slot(*t.tab, key)) { *slot = val; return; }
        if (key.type == Type::Number && key.num == double(t.tab->kv.size() + 1) &&
            (t.tab->kv.empty() || array_slot(*t.tab, t.tab->kv.back().first))) { t.tab->kv.push_back({key, val}); return; }   // append to the array part
        for (auto& kv : t.tab->kv)
            if (key_equal(kv.first, key)) { kv.second = val; return; }
        t.tab->kv.push_back({key, val});
    }
};

} // namespace fakelua
