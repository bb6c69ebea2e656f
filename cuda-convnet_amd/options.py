"""Command-line options for the convnet driver (Python 3).

Behavioural mirror of the reference's Python/options.py:79-178 (OptionsParser over getopt:
mandatory options unless a default or an excusing option is given, `requires`, `set_once`,
expression defaults evaluated over the other options, merge of command-line values over the
options saved in a checkpoint).  Class names are kept because checkpoints pickle the
OptionsParser object under these names (Python/gpumodel.py:272-273).
"""
import os
import re
import sys
from getopt import getopt

_BOLD, _PLAIN = "\033[1m", "\033[0m"


class OptionException(Exception):
    pass


class OptionMissingException(OptionException):
    pass


class OptionExpression(object):
    """A default given as a Python expression over the other options' values,
    e.g. OptionExpression("[-1] * num_gpus") (Python/gpumodel.py:313)."""

    def __init__(self, expr):
        self.expr = expr

    def evaluate(self, env):
        try:
            return eval(self.expr, {}, dict(env))
        except Exception as e:
            raise OptionException("expression '%s': unable to parse: %s" % (self.expr, e))


# ---------------------------------------------------------------- value parsers
class OptionParser(object):
    type_str = None

    @staticmethod
    def parse(value):
        return str(value)

    @staticmethod
    def to_string(value):
        return str(value)

    @classmethod
    def get_type_str(cls):
        return cls.type_str


class StringOptionParser(OptionParser):
    type_str = "string"

    @staticmethod
    def is_type(value):
        return isinstance(value, str)


class IntegerOptionParser(OptionParser):
    type_str = "int"

    @staticmethod
    def parse(value):
        try:
            return int(value)
        except (TypeError, ValueError):
            raise OptionException("argument is not an integer")

    @staticmethod
    def is_type(value):
        return type(value) == int


class BooleanOptionParser(OptionParser):
    type_str = "0/1"

    @staticmethod
    def parse(value):
        try:
            v = int(value)
        except (TypeError, ValueError):
            v = None
        if v not in (0, 1):
            raise OptionException("argument is not a boolean")
        return v

    @staticmethod
    def is_type(value):
        return type(value) == int and value in (0, 1)


class FloatOptionParser(OptionParser):
    type_str = "float"

    @staticmethod
    def parse(value):
        try:
            return float(value)
        except (TypeError, ValueError):
            raise OptionException("argument is not a float")

    @staticmethod
    def to_string(value):
        return "%.6g" % value

    @staticmethod
    def is_type(value):
        return type(value) == float


class RangeOptionParser(OptionParser):
    """`a-b` (inclusive) or a single integer -> list of ints (Python/options.py:328-336)."""
    type_str = "int[-int]"

    @staticmethod
    def parse(value):
        m = re.match(r"^(\d+)\-(\d+)$", value)
        try:
            if m:
                return list(range(int(m.group(1)), int(m.group(2)) + 1))
            return [int(value)]
        except (TypeError, ValueError):
            raise OptionException("argument is neither an integer nor a range")

    @staticmethod
    def to_string(value):
        return "%d-%d" % (value[0], value[-1])

    @staticmethod
    def is_type(value):
        return isinstance(value, list)


class ListOptionParser(OptionParser):
    """Separator-delimited list; `parsers` is one parser (any length) or a list (fixed form)."""

    def __init__(self, parsers, sepchar=","):
        self.parsers = parsers
        self.sepchar = sepchar

    def _pairs(self, values):
        if isinstance(self.parsers, list):
            return zip(self.parsers, values)
        return ((self.parsers, v) for v in values)

    def parse(self, value):
        values = value.split(self.sepchar)
        if isinstance(self.parsers, list) and len(values) != len(self.parsers):
            raise OptionException("requires %d arguments, given %d" % (len(self.parsers), len(values)))
        try:
            return [p.parse(v) for p, v in self._pairs(values)]
        except OptionException:
            raise OptionException("argument is not of the form %s" % self.get_type_str())

    def to_string(self, value):
        return self.sepchar.join(p.to_string(v) for p, v in self._pairs(value))

    def get_type_str(self):
        if isinstance(self.parsers, list):
            return self.sepchar.join(p.get_type_str() for p in self.parsers)
        return "%s%s..." % (self.parsers.get_type_str(), self.sepchar)

    @staticmethod
    def is_type(value):
        return isinstance(value, list)


# ---------------------------------------------------------------- one option
class Option(object):
    def __init__(self, letter, name, desc, parser, set_once, default, excuses, requires, save):
        assert name is not None
        self.letter, self.name, self.desc, self.parser = letter, name, desc, parser
        self.set_once, self.default, self.excuses, self.requires, self.save = set_once, default, excuses, requires, save
        self.value = None
        self.value_given = False
        # single-letter options take one dash, the others two (Python/options.py:49)
        self.prefixed_letter = min(2, len(letter)) * "-" + letter

    def set_value(self, value, parse=True):
        try:
            self.value = self.parser.parse(value) if parse else value
            self.value_given = True
        except OptionException as e:
            raise OptionException("Unable to parse option %s (%s): %s" % (self.prefixed_letter, self.desc, e))

    def set_default(self):
        if self.default is not None:
            self.value = self.default

    def eval_expr_default(self, env):
        if not isinstance(self.default, OptionExpression) or self.value_given:
            return
        try:
            self.value = self.default.evaluate(env)
            if not self.parser.is_type(self.value):
                raise OptionException("expression result %s is not of right type (%s)" % (self.value, self.parser.get_type_str()))
        except Exception as e:
            raise OptionException("Unable to set default value for option %s (%s): %s" % (self.prefixed_letter, self.desc, e))

    def get_str_value(self, get_default_str=False):
        val = self.default if get_default_str else self.value
        if val is None:
            return ""
        if isinstance(val, OptionExpression):
            return val.expr
        return self.parser.to_string(val)


# ---------------------------------------------------------------- the option set
class OptionsParser(object):
    SORT_LETTER, SORT_DESC, SORT_EXPR_LAST = 1, 2, 3
    EXCLUDE_ALL = "all"

    def __init__(self):
        self.options = {}

    def add_option(self, letter, name, parser, desc, set_once=False, default=None, excuses=(), requires=(), save=True):
        assert name not in self.options
        self.options[name] = Option(letter, name, desc, parser, set_once, default, excuses, requires, save)

    def set_value(self, name, value, parse=True):
        self.options[name].set_value(value, parse=parse)

    def get_value(self, name):
        return self.options[name].value

    def delete_option(self, name):
        self.options.pop(name, None)

    def _excused(self, opt, given):
        return any(o.prefixed_letter in given and (o.excuses == self.EXCLUDE_ALL or opt.name in o.excuses)
                   for o in self.options.values())

    def parse(self, eval_expr_defaults=False, argv=None):
        argv = sys.argv[1:] if argv is None else argv
        shorts = "".join("%s:" % o.letter for o in self.options.values() if len(o.letter) == 1)
        longs = ["%s=" % o.letter for o in self.options.values() if len(o.letter) > 1]
        given = dict(getopt(argv, shorts, longopts=longs)[0])
        for o in self.get_options_list(sort_order=self.SORT_EXPR_LAST):
            if o.prefixed_letter in given:
                o.set_value(given[o.prefixed_letter])
                for need in self.get_options_list(sort_order=self.SORT_LETTER):
                    if need.name in o.requires and need.prefixed_letter not in given:
                        raise OptionMissingException("Option %s (%s) requires option %s (%s)" % (
                            o.prefixed_letter, o.desc, need.prefixed_letter, need.desc))
            else:
                if o.default is None and not self._excused(o, given):
                    raise OptionMissingException("Option %s (%s) not supplied" % (o.prefixed_letter, o.desc))
                o.set_default()
        if eval_expr_defaults:
            self.eval_expr_defaults()
        return self.options

    def merge_from(self, op2):
        """Take op2's explicitly given (or non-saved) options over this (loaded) set; options this
        set lacks are added (Python/options.py:160-170)."""
        for name in list(self.options):
            o2 = op2.options.get(name)
            if o2 is not None and (o2.value_given or not o2.save):
                if o2.set_once:
                    raise OptionException("Option %s (%s) cannot be changed" % (o2.prefixed_letter, o2.desc))
                self.options[name] = o2
        for name, o2 in op2.options.items():
            self.options.setdefault(name, o2)

    def eval_expr_defaults(self):
        env = dict((name, o.value) for name, o in self.options.items())
        for o in self.options.values():
            o.eval_expr_default(env)

    def all_values_given(self):
        return max(o.value_given for o in self.options.values() if o.default is not None)

    def get_options_list(self, sort_order=SORT_LETTER):
        opts = list(self.options.values())
        if sort_order == self.SORT_LETTER:
            return sorted(opts, key=lambda o: o.letter)
        if sort_order == self.SORT_EXPR_LAST:
            return sorted(opts, key=lambda o: isinstance(o.default, OptionExpression))
        return sorted(opts, key=lambda o: o.desc)

    def print_usage(self, print_constraints=False):
        print("%s usage:" % os.path.basename(sys.argv[0]))
        rows_opt, rows_mand = [], []
        for o in self.get_options_list():
            exc = " "
            if o.default is None:
                exc = ", ".join(sorted(o2.prefixed_letter for o2 in self.options.values()
                                       if o2.excuses == self.EXCLUDE_ALL or o.name in o2.excuses))
            req = ", ".join(sorted(o2.prefixed_letter for o2 in self.options.values() if o2.name in o.requires))
            row = (self._bold(o.prefixed_letter) + " <%s>" % o.parser.get_type_str(), o.desc,
                   None if o.default is None else "[%s]" % o.get_str_value(get_default_str=True), exc, req)
            (rows_mand if o.default is None else rows_opt).append(row)
        rows = rows_opt + rows_mand
        widths = [self._longest_value(rows, key=lambda r, i=i: r[i]) for i in range(4)]
        heads = ["    Option", "Description", "Default"] + (["Excused by", "Requires"] if print_constraints else [])
        print(" ".join(self._bold(h.ljust(widths[i])) for i, h in enumerate(heads)))
        for l, d, de, ex, req in rows:
            lead = ("     %s  -" if de is None else "    [%s] -") % l.ljust(widths[0])
            tail = (" %s %s" % (ex.ljust(widths[3]), req)) if print_constraints else ""
            print("%s %s %s%s" % (lead, d.ljust(widths[1]), (de or " ").ljust(widths[2]), tail))

    def print_values(self):
        wd = self._longest_value(self.options.values(), key=lambda o: o.desc)
        defaulted = [o for o in self.options.values() if not o.value_given and o.default is not None]
        wv = self._longest_value(defaulted, key=lambda o: o.get_str_value()) if defaulted else 0
        for o in self.get_options_list(sort_order=self.SORT_DESC):
            mark = "[DEFAULT]" if (not o.value_given and o.default is not None) else ""
            print("%s: %s %s" % (o.desc.ljust(wd), o.get_str_value().ljust(wv), mark))

    @staticmethod
    def _longest_value(values, key=lambda x: x):
        return max([0] + [len(key(v)) for v in values if key(v) is not None])

    @staticmethod
    def _bold(s):
        return _BOLD + s + _PLAIN
