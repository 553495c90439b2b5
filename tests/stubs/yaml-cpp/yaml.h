// declaration-only stand-in (see tests/stubs/README.md)
#ifndef EBVO_STUB_YAML_H
#define EBVO_STUB_YAML_H
#include <string>
#include <vector>
namespace YAML
{
struct Node
{
    Node();
    template <class K> Node operator[](const K &) const;
    template <class T> T as() const;
    template <class T> T as(const T &) const;
    bool IsDefined() const;
    bool IsNull() const;
    bool IsSequence() const;
    bool IsMap() const;
    size_t size() const;
    explicit operator bool() const;
    struct iterator
    {
        Node operator*() const;
        iterator &operator++();
        bool operator!=(const iterator &) const;
    };
    iterator begin() const;
    iterator end() const;
};
Node LoadFile(const std::string &);
Node Load(const std::string &);
struct Exception : std::exception
{
    const char *what() const noexcept override;
};
typedef Exception BadFile;
} // namespace YAML
#endif
