// stand-in for the compile check of include/cvo_adaptor.hpp (tests/stubs/README.md): declarations only
#pragma once
#include <string>
namespace cv {
struct Mat { unsigned char* data; int rows, cols; bool isContinuous() const; Mat clone() const; };
struct Point2f { float x, y; Point2f(); Point2f(float, float); };
struct FileNode { operator float() const; operator int() const; };
struct FileStorage { enum { READ = 0 }; FileStorage(const std::string&, int); FileNode operator[](const char*) const; };
}
