// LuaShim.h -- the slice of GarrysMod::Lua::ILuaBase (gmod-module-base, an un-vendored
// submodule of the reference) that the Tracing API of VisTrace uses
// (source/VisTrace.cpp:749-843 and 457-747, source/objects/AccelStruct.cpp:533-838).
// In a real build of the module this header is replaced by "GarrysMod/Lua/Interface.h";
// method names, argument order and the non-returning behaviour of ThrowError / ArgError /
// CheckType follow that interface so the host classes compile against either.
#pragma once

#include <cstddef>

struct Vector { float x, y, z; };   // Source SDK Vector as the Lua interface hands it out

namespace GarrysMod { namespace Lua {

namespace Type {
enum : int { None = -1, Nil = 0, Bool = 1, LightUserData = 2, Number = 3, String = 4, Table = 5, Function = 6,
             UserData = 7, Thread = 8, Entity = 9, Vector = 10, Angle = 11, Count = 44 /* first free user type */ };
}

// ILuaBase::PushSpecial arguments
enum { SPECIAL_GLOB = 0, SPECIAL_ENV = 1, SPECIAL_REG = 2 };

class ILuaBase;
typedef int (*CFunc)(ILuaBase* LUA);

class ILuaBase {
public:
    virtual ~ILuaBase() {}
    virtual int          Top() = 0;
    virtual void         Push(int stackPos) = 0;
    virtual void         Pop(int amount = 1) = 0;
    virtual void         CreateTable() = 0;
    virtual void         PushSpecial(int type) = 0;                       // SPECIAL_GLOB: the global table
    virtual void         GetField(int stackPos, const char* name) = 0;    // pushes t[name]
    virtual void         GetTable(int stackPos) = 0;                      // pops a key, pushes t[key]
    virtual void         Call(int nargs, int nresults) = 0;               // pops function + args, pushes the results
    virtual void         SetField(int stackPos, const char* name) = 0;    // t[name] = top; pops value
    virtual void         SetTable(int stackPos) = 0;                      // t[key] = value; pops both
    virtual void         SetMetaTable(int stackPos) = 0;                  // pops the metatable
    virtual int          Next(int stackPos) = 0;                         // lua_next
    [[noreturn]] virtual void ThrowError(const char* msg) = 0;
    virtual void         CheckType(int stackPos, int type) = 0;          // throws a formatted error on mismatch
    [[noreturn]] virtual void ArgError(int argNum, const char* msg) = 0;
    virtual bool         IsType(int stackPos, int type) = 0;
    virtual int          GetType(int stackPos) = 0;
    virtual double       GetNumber(int stackPos = -1) = 0;
    virtual double       CheckNumber(int stackPos = -1) = 0;
    virtual bool         GetBool(int stackPos = -1) = 0;
    virtual const char*  GetString(int stackPos = -1, unsigned int* outLen = nullptr) = 0;   // binary safe: length in *outLen
    virtual const Vector& GetVector(int stackPos = -1) = 0;
    virtual void         PushNil() = 0;
    virtual void         PushNumber(double v) = 0;
    virtual void         PushBool(bool v) = 0;
    virtual void         PushString(const char* s, unsigned len = 0) = 0;
    virtual void         PushVector(const Vector& v) = 0;
    virtual void         PushCFunction(CFunc f) = 0;
    virtual int          CreateMetaTable(const char* name) = 0;          // pushes it, returns its type id
    virtual void         PushUserType(void* data, int type) = 0;
    virtual void         SetUserType(int stackPos, void* data) = 0;
    virtual void*        GetUserdataRaw(int stackPos, int type) = 0;     // pointer stored by PushUserType

    template <class T> T* GetUserType(int stackPos, int type) { return static_cast<T*>(GetUserdataRaw(stackPos, type)); }
    // the real interface boxes a copy of `value`; here `value` is always a pointer, stored as is
    template <class T> void PushUserType_Value(T* value, int type) { PushUserType(static_cast<void*>(value), type); }
};

} } // namespace GarrysMod::Lua

#define LUA_FUNCTION(NAME) int NAME(GarrysMod::Lua::ILuaBase* LUA)
