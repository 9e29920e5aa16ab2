// tool.hpp -- minimal re-creation of the gatb-core "Tool" CLI framework, limited
// to what the DSK sources touch:
//   src/DSK.hpp:39            class DSK : public Tool
//   src/DSK.cpp:80-87         getParser()->push_back(...), getParser(STR_URI_INPUT)->setName(STR_URI_FILE)
//   src/DSK.cpp:51,57,100     getInput()->getStr/getInt/add
//   src/DSK.cpp:63-68         getInfo()->add(depth, props), getXML()
//   src/main.cpp:34-46        Tool::run, OptionFailure::displayErrors, Exception::getMessage
//   utils/dsk2ascii.cpp:16-22 OptionOneParam / OptionNoParam / push_front / saw
// Progress bars, XML readers, observers etc. of gatb-core are out of scope.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#include <type_traits>

namespace dsk {

#define STR_URI_INPUT "-in"
#define STR_URI_FILE "-file"
#define STR_URI_OUTPUT "-out"
#define STR_URI_OUTPUT_DIR "-out-dir"
#define STR_URI_OUTPUT_TMP "-out-tmp"
#define STR_KMER_SIZE "-kmer-size"
#define STR_KMER_ABUNDANCE_MIN "-abundance-min"
#define STR_KMER_ABUNDANCE_MAX "-abundance-max"
#define STR_HISTOGRAM_MAX "-histo-max"
#define STR_VERBOSE "-verbose"
#define STR_NB_CORES "-nb-cores"
#define STR_MAX_MEMORY "-max-memory"
#define STR_MAX_DISK "-max-disk"
#define STR_HELP "-help"
#define STR_VERSION "-version"

class Exception {
public:
    Exception() {}
    explicit Exception(const char* fmt, ...) {
        char buf[2048]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); msg_ = buf;
    }
    explicit Exception(const std::string& m) : msg_(m) {}
    const char* getMessage() const { return msg_.c_str(); }
private:
    std::string msg_;
};

// ---------------------------------------------------------------- properties
class IProperties {
public:
    struct Entry { size_t depth; std::string key, value; };
    void add(size_t depth, const std::string& key, const std::string& value = "") { entries_.push_back({depth, key, value}); }
    void add(size_t depth, const std::string& key, const char* fmt, ...) {
        char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
        entries_.push_back({depth, key, buf});
    }
    void add(size_t depth, const IProperties* other) { if (other) for (auto& e : other->entries_) entries_.push_back({depth + e.depth, e.key, e.value}); }
    void add(size_t depth, const IProperties& other) { add(depth, &other); }
    // set-or-replace at depth 0 (what the option parser does)
    void set(const std::string& key, const std::string& value) {
        for (auto& e : entries_) if (e.key == key) { e.value = value; return; }
        entries_.push_back({0, key, value});
    }
    bool has(const std::string& key) const { for (auto& e : entries_) if (e.key == key) return true; return false; }
    std::string getStr(const std::string& key) const {
        for (auto& e : entries_) if (e.key == key) return e.value;
        throw Exception("Empty property for key '%s'", key.c_str());
    }
    long long getInt(const std::string& key) const { return atoll(getStr(key).c_str()); }
    double getDouble(const std::string& key) const { return atof(getStr(key).c_str()); }
    const std::vector<Entry>& entries() const { return entries_; }
    // nested XML by depth, like the "xml" attribute stored at src/DSK.cpp:68
    std::string getXML() const {
        std::ostringstream os; std::vector<std::string> open;
        for (size_t i = 0; i < entries_.size(); ++i) {
            const Entry& e = entries_[i];
            while (open.size() > e.depth) { os << std::string(open.size() - 1, ' ') << "</" << open.back() << ">\n"; open.pop_back(); }
            bool parent = i + 1 < entries_.size() && entries_[i + 1].depth > e.depth;
            std::string tag = tagOf(e.key);
            os << std::string(e.depth, ' ') << "<" << tag << ">" << e.value;
            if (parent) { os << "\n"; open.push_back(tag); }
            else os << "</" << tag << ">\n";
        }
        while (!open.empty()) { os << std::string(open.size() - 1, ' ') << "</" << open.back() << ">\n"; open.pop_back(); }
        return os.str();
    }
    void dump(std::ostream& os) const { for (auto& e : entries_) os << std::string(4 * e.depth, ' ') << e.key << (e.value.empty() ? "" : " : ") << e.value << "\n"; }
private:
    static std::string tagOf(const std::string& k) { std::string t; for (char c : k) t += (isalnum((unsigned char)c) || c == '_') ? c : '_'; if (t.empty() || isdigit((unsigned char)t[0])) t = "_" + t; return t; }
    std::vector<Entry> entries_;
};

// ---------------------------------------------------------------- options
class OptionFailure {
public:
    OptionFailure(const std::string& toolName, const std::string& usage, const std::vector<std::string>& errors, bool helpOnly = false)
        : tool_(toolName), usage_(usage), errors_(errors), help_(helpOnly) {}
    int displayErrors(std::ostream& os) const {
        for (auto& e : errors_) os << "ERROR: " << e << "\n";
        os << usage_;
        return help_ ? EXIT_SUCCESS : EXIT_FAILURE;
    }
private:
    std::string tool_, usage_; std::vector<std::string> errors_; bool help_;
};

class IOptionsParser {
public:
    explicit IOptionsParser(const std::string& name, const std::string& help = "") : name_(name), help_(help) {}
    virtual ~IOptionsParser() {}
    const std::string& getName() const { return name_; }
    void setName(const std::string& n) { name_ = n; }
    const std::string& getHelp() const { return help_; }
    virtual int nbArgs() const { return -1; }          // -1: composite
    virtual bool mandatory() const { return false; }
    virtual std::string defaultValue() const { return ""; }
    virtual bool hasDefault() const { return false; }
    bool visible = true;
private:
    std::string name_, help_;
};

class OptionNoParam : public IOptionsParser {
public:
    OptionNoParam(const std::string& name, const std::string& help, bool mandatory = false) : IOptionsParser(name, help), mand_(mandatory) {}
    int nbArgs() const override { return 0; }
    bool mandatory() const override { return mand_; }
private:
    bool mand_;
};

class OptionOneParam : public IOptionsParser {
public:
    OptionOneParam(const std::string& name, const std::string& help, bool mandatory = false)
        : IOptionsParser(name, help), mand_(mandatory), hasDef_(false) {}
    OptionOneParam(const std::string& name, const std::string& help, bool mandatory, const std::string& def, bool vis = true)
        : IOptionsParser(name, help), mand_(mandatory), hasDef_(true), def_(def) { visible = vis; }
    int nbArgs() const override { return 1; }
    bool mandatory() const override { return mand_; }
    bool hasDefault() const override { return hasDef_; }
    std::string defaultValue() const override { return def_; }
private:
    bool mand_, hasDef_; std::string def_;
};

class OptionsParser : public IOptionsParser {
public:
    explicit OptionsParser(const std::string& name, const std::string& help = "") : IOptionsParser(name, help) {}
    ~OptionsParser() override { for (auto* c : children_) delete c; }
    void push_back(IOptionsParser* p, size_t /*expandDepth*/ = 0) { if (p) children_.push_back(p); }
    void push_front(IOptionsParser* p, size_t /*expandDepth*/ = 0) { if (p) children_.push_front(p); }
    // recursive lookup by option / sub-parser name (src/DSK.cpp:86)
    IOptionsParser* getParser(const std::string& name) {
        if (getName() == name) return this;
        for (auto* c : children_) {
            if (c->getName() == name) return c;
            if (auto* op = dynamic_cast<OptionsParser*>(c)) if (auto* r = op->getParser(name)) return r;
        }
        return nullptr;
    }
    bool saw(const std::string& name) const { return seen_.count(name) != 0; }
    void collect(std::vector<IOptionsParser*>& out) {
        for (auto* c : children_) { if (auto* op = dynamic_cast<OptionsParser*>(c)) op->collect(out); else out.push_back(c); }
    }
    std::string usage(const std::string& toolName) {
        std::vector<IOptionsParser*> opts; collect(opts);
        std::ostringstream os;
        os << "\n[" << toolName << " options]\n";
        for (auto* o : opts) {
            if (!o->visible) continue;
            char line[512];
            std::string left = o->getName() + (o->nbArgs() == 1 ? " (1 arg)" : " (0 arg)");
            std::string def = o->hasDefault() ? ("  [default '" + o->defaultValue() + "']") : "";
            snprintf(line, sizeof(line), "       %-28s :    %s%s\n", left.c_str(), o->getHelp().c_str(), def.c_str());
            os << line;
        }
        return os.str();
    }
    // Parse argv into properties; throws OptionFailure on unknown/missing options.
    IProperties* parse(int argc, char** argv, const std::string& toolName) {
        std::vector<IOptionsParser*> opts; collect(opts);
        std::map<std::string, IOptionsParser*> byName;
        for (auto* o : opts) byName[o->getName()] = o;
        auto props = std::unique_ptr<IProperties>(new IProperties());
        std::vector<std::string> errors; bool help = false;
        seen_.clear();
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            auto it = byName.find(a);
            if (it == byName.end()) { errors.push_back("Unknown parameter '" + a + "'"); continue; }
            seen_.insert(a);
            if (a == STR_HELP) help = true;
            if (it->second->nbArgs() == 1) {
                if (i + 1 >= argc) { errors.push_back("Too few arguments for the " + a + " option..."); continue; }
                props->set(a, argv[++i]);
            } else props->set(a, "");
        }
        if (help) throw OptionFailure(toolName, usage(toolName), {}, true);
        for (auto* o : opts) {
            if (props->has(o->getName())) continue;
            if (o->mandatory()) errors.push_back("Option '" + o->getName() + "' is mandatory");
            else if (o->hasDefault()) props->set(o->getName(), o->defaultValue());
        }
        if (!errors.empty()) throw OptionFailure(toolName, usage(toolName), errors);
        return props.release();
    }
private:
    std::list<IOptionsParser*> children_;
    std::set<std::string> seen_;
};

// ---------------------------------------------------------------- Tool
// Tool::run = parse argv -> getInput() properties -> execute() -> print the
// info tree when -verbose > 0 (src/main.cpp:34, src/DSK.cpp:57,63-64).
template <class T> class Iterator;      // storage.hpp

class Tool {
public:
    explicit Tool(const std::string& name) : name_(name), parser_(new OptionsParser(name)), input_(nullptr), info_(new IProperties()) {
        parser_->push_back(new OptionOneParam(STR_NB_CORES, "number of cores", false, "0"));
        parser_->push_back(new OptionOneParam(STR_VERBOSE, "verbosity level", false, "1"));
        parser_->push_back(new OptionNoParam(STR_VERSION, "version", false));
        parser_->push_back(new OptionNoParam(STR_HELP, "help", false));
    }
    virtual ~Tool() { delete parser_; delete input_; delete info_; }
    const std::string& getName() const { return name_; }
    OptionsParser* getParser() { return parser_; }
    IProperties* getInput() { return input_; }
    IProperties* getInfo() { return info_; }
    // `tool.createIterator(collection.iterator(), collection.getNbItems(), "parsing")` (utils/dsk2ascii.cpp:77): gatb-core
    // wraps the iterator in a progress notifier when -verbose asks for one; here it is handed through unchanged (the
    // caller owns it, e.g. with LOCAL), and the count and the message are accepted for source compatibility.
    template <class T>
    Iterator<T>* createIterator(Iterator<T>* iter, size_t nbIterations = 0, const char* message = nullptr) {
        (void)nbIterations; (void)message;
        return iter;
    }
    virtual std::string getVersion() const { return "dsk_amd 0.1 (MI355X-native count path; CLI surface of DSK 2.3.1)"; }

    IProperties* run(int argc, char** argv) {
        delete input_; input_ = nullptr;
        for (int i = 1; i < argc; ++i) if (std::string(argv[i]) == STR_VERSION) { std::cout << name_ << " " << getVersion() << std::endl; return nullptr; }
        input_ = parser_->parse(argc, argv, name_);
        info_->add(0, name_);
        execute();
        if (input_->has(STR_VERBOSE) && input_->getInt(STR_VERBOSE) > 0) info_->dump(std::cout);
        return input_;
    }
protected:
    virtual void execute() = 0;
private:
    std::string name_;
    OptionsParser* parser_;
    IProperties* input_;
    IProperties* info_;
};

// LOCAL(x): scope-bound ownership in gatb-core (src/DSK.cpp:52); here objects are
// plain heap objects released by a guard.
template <class T> struct LocalGuard { T* p; explicit LocalGuard(T* q) : p(q) {} ~LocalGuard() { delete p; } };
#define DSK_CAT2(a, b) a##b
#define DSK_CAT(a, b) DSK_CAT2(a, b)
#define LOCAL(x) ::dsk::LocalGuard<typename std::remove_pointer<decltype(x)>::type> DSK_CAT(local_guard_, __LINE__)(x)

}  // namespace dsk
