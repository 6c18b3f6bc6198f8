// UnrealHeaderTool would generate this file; the syntax check needs only `Super` inside AOctreeSearch (see README.md).
#pragma once
#undef GENERATED_BODY
#define GENERATED_BODY(...) public: typedef AActor Super; private:
